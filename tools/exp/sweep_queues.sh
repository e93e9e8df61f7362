# usage (GPU box): QUEUES="4 5 8" bash tools/exp/sweep_queues.sh [bench flags]  -> ms/step per GPU_MAX_HW_QUEUES value
R=$GRAFT_REPO_ROOT
for q in ${QUEUES:-4 2 3 5 6 8}; do
  GPU_MAX_HW_QUEUES=$q python $R/bench.py --no-cpu-baseline --rotate-batches 0 --no-fg-capped --no-device-targets --no-direct-head-forward --pipeline-examples 0 "$@" 2>/dev/null > /tmp/b.json
  python -c "import json; d=json.load(open('/tmp/b.json')); print('queues', $q, d['value'], d['ms_per_step'])"
done
