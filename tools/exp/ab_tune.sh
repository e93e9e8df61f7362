# usage (GPU box): bash tools/exp/ab_tune.sh "<knob=value>" ["<knob=value>" ...] -> headline region per setting, twice, alternating
Q="--no-cpu-baseline --rotate-batches 0 --no-fg-capped --no-direct-head-forward --no-device-targets --pipeline-examples 0 --no-extra-workloads --no-fp32-mfma --repeats 3"
for rep in 1 2; do for t in "$@"; do echo "== $t"; python bench.py $Q --tune "$t" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['repeats']['ms_per_step'], d['config']['loss'])"; done; done
