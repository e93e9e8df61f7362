# usage (GPU box): bash tools/exp/wino_res4_ab.sh -> headline ms/step with res4's 3x3 layers on the Winograd route (work threshold 2^25) vs shipped (2^27)
cd $GRAFT_REPO_ROOT
ONLY="--no-cpu-baseline --rotate-batches 0 --no-fg-capped --no-device-targets --no-direct-head-forward --no-fp32-mfma --pipeline-examples 0 --no-extra-workloads --no-profile"
for rep in 1 2 3; do
  for v in 134217728 33554432; do
    r=$(MRCNN_WINO_MIN_WORK=$v python bench.py $ONLY 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['repeats']['ms_per_step'], d['config']['loss'])")
    echo "MRCNN_WINO_MIN_WORK=$v: $r"
  done
done
