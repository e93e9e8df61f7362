# usage (GPU box): bash tools/exp/tail_sweep.sh -> bench_conv on the backbone shapes with / without the K-split of leftover rows
cd $GRAFT_REPO_ROOT
for t in "small_rem_max=154" "small_rem_max=0" "fused_tail=0"; do
  echo "== $t"
  for s in "res4 3x3" "res4 1x1 256" "res4 1x1 1024" "res3 3x3" "res3 1x1 128" "res3 1x1 512"; do
    BENCH_TUNE=$t python tools/bench_conv.py "$s" 2>/dev/null | grep -v "^shape\|^sum"
  done
done
