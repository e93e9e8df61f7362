# usage (GPU box): bash tools/exp/sweep_defer.sh -> gpurun_out/sweep_defer.txt (headline region only per setting)
Q="--no-cpu-baseline --rotate-batches 0 --no-fg-capped --no-direct-head-forward --no-device-targets --pipeline-examples 0 --no-extra-workloads --no-fp32-mfma --repeats 3"
out=gpurun_out/sweep_defer.txt; : > $out
for d in 5 3 4 6 7 8 0 5; do
  echo "== defer $d" >> $out
  python bench.py $Q --defer-wgrad $d 2>>gpurun_out/sweep_defer.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['repeats']['ms_per_step'])" >> $out
done
cat $out
