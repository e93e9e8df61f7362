#!/bin/bash
# Same-box A/B of projected pooling (functions/conv.py): headline region only, alternating arms.
# usage: tools/exp/ab_projected.sh [extra bench.py flags]; writes gpurun_out/ab_projected.txt
mkdir -p gpurun_out
out=gpurun_out/ab_projected.txt
: > $out
Q="--no-cpu-baseline --rotate-batches 0 --no-fg-capped --no-direct-head-forward --no-device-targets --pipeline-examples 0 --no-extra-workloads --no-fp32-mfma --repeats 3"
for rep in 1 2; do
  for arm in 0 1; do
    echo "== MRCNN_PROJECTED_POOLING=$arm rep $rep $*" >> $out
    MRCNN_PROJECTED_POOLING=$arm python bench.py $Q "$@" 2>>gpurun_out/ab_projected.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d.get('repeats'), d['roofline'].get('achieved'), d['config'].get('loss'))" >> $out
  done
done
cat $out
