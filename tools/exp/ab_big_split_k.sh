Q="--no-cpu-baseline --rotate-batches 0 --no-fg-capped --no-direct-head-forward --no-device-targets --pipeline-examples 0 --no-extra-workloads --no-fp32-mfma --repeats 3"
for rep in 1 2; do for t in big_split_k=0 big_split_k=-1; do for L in 101 50; do echo "== R$L $t"; python bench.py $Q --layers $L --tune "$t" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['repeats']['ms_per_step'])"; done; done; done
