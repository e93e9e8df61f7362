# usage (GPU box): bash tools/exp/ab_bench.sh "<bench flags A>" "<bench flags B>" [rounds] -> alternating headline runs on ONE box
A="$1"; B="$2"; N=${3:-3}
cd $GRAFT_REPO_ROOT
ONLY="--no-cpu-baseline --rotate-batches 0 --no-fg-capped --no-device-targets --no-direct-head-forward --no-fp32-mfma --pipeline-examples 0 --no-extra-workloads --no-profile"
for i in $(seq 1 $N); do
  for v in A B; do
    if [ $v = A ]; then F="$A"; else F="$B"; fi
    python bench.py $ONLY $F 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', '[$F]', d['ms_per_step'], d['repeats']['ms_per_step'])"
  done
done
