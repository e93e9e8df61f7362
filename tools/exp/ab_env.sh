# usage (GPU box): bash tools/exp/ab_env.sh "<ENV=VAL ...>" [bench flags]   -> ms/step, A/B/A/B
R=$GRAFT_REPO_ROOT
E="$1"; shift
for i in 1 2; do
  for v in "" "$E"; do
    env $v python $R/bench.py --no-cpu-baseline --rotate-batches 0 --no-fg-capped --no-device-targets --no-direct-head-forward --pipeline-examples 0 "$@" 2>/dev/null > /tmp/b.json
    python -c "import json; d=json.load(open('/tmp/b.json')); print('[%s]' % '$v', d['value'], d['ms_per_step'], d['config']['loss'])"
  done
done
