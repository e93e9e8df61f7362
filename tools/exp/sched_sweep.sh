# usage (GPU box): bash tools/exp/sched_sweep.sh <tag> -> headline ms/step under the scheduling knobs (same box, interleaved twice)
TAG=${1:-r05}
cd $GRAFT_REPO_ROOT
ONLY="--no-cpu-baseline --rotate-batches 0 --no-fg-capped --no-device-targets --no-direct-head-forward --no-fp32-mfma --pipeline-examples 0 --no-extra-workloads --no-profile"
OUT=gpurun_out/${TAG}_sched_sweep.txt
: > $OUT
run() { # label, env, flags
  r=$(env $2 python bench.py $ONLY $3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['repeats']['ms_per_step'])")
  echo "$1: $r" | tee -a $OUT
}
for rep in 1 2; do
  run "default (side stream for small wgrads, defer 5)" "X=1" ""
  run "no wgrad side stream" "MRCNN_SIDE_WGRAD_MAX_PIXELS=0" ""
  run "side stream for ALL wgrads" "MRCNN_SIDE_WGRAD_MAX_PIXELS=100000000" ""
  run "defer 0" "X=1" "--defer-wgrad 0"
  run "defer 3" "X=1" "--defer-wgrad 3"
  run "defer 7" "X=1" "--defer-wgrad 7"
  run "defer 10" "X=1" "--defer-wgrad 10"
  run "no frozen-prefix prefetch" "X=1" "--no-prefetch-frozen"
done
