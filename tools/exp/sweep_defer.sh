# usage (GPU box): bash tools/exp/sweep_defer.sh  -> ms/step for several --defer-wgrad counts
R=$GRAFT_REPO_ROOT
for n in 5 0 3 4 6 7 8 10 5; do
  python $R/bench.py --no-cpu-baseline --rotate-batches 0 --no-fg-capped --no-device-targets --no-direct-head-forward --pipeline-examples 0 --defer-wgrad $n 2>/dev/null > /tmp/b.json
  python -c "import json; d=json.load(open('/tmp/b.json')); print('defer', $n, d['value'], d['ms_per_step'])"
done
