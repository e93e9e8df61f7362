R=$GRAFT_REPO_ROOT
Q="--no-cpu-baseline --rotate-batches 0 --no-fg-capped --no-direct-head-forward --no-device-targets --pipeline-examples 0 --no-extra-workloads --no-fp32-mfma --repeats 3"
python -m pytest tests/test_gpu_conv.py -q -k "w8" -x 2>&1 | tail -2
for rep in 1 2; do for v in head base; do
  if [ "$v" = base ]; then unset MRCNN_HIP_LIB; else export MRCNN_HIP_LIB=$R/chainer_mask_rcnn_amd/csrc/variants/lib$v.so; fi
  echo "== $v"
  for s in "res5 1x1" "res5 3x3"; do python tools/bench_conv.py "$s" 2>/dev/null | grep -v "^shape\|^sum"; done
  python bench.py $Q 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['repeats']['ms_per_step'], d['config']['loss'], d['roofline']['achieved'])"
done; done
