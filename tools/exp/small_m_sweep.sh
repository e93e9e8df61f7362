# usage (GPU box): bash tools/exp/small_m_sweep.sh <tag> -> bench_conv on the backbone shapes under the small-M tile-policy knobs
TAG=${1:-r05}
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${TAG}_small_m.txt
: > $OUT
for t in "small_m_split=0" "small_m_split=2" "small_m_split=3" "small_m_split=4" "small_m_split=6" "small_m_split=8"; do
  echo "== $t" >> $OUT
  for s in "res4 3x3" "res4 1x1 256" "res4 1x1 1024" "res3 3x3" "res3 1x1 128" "res3 1x1 512" "res2 3x3" "stem-like"; do
    BENCH_TUNE=$t python tools/bench_conv.py "$s" 2>/dev/null | grep -v "^shape\|^sum" >> $OUT
  done
done
cat $OUT
