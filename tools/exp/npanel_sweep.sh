# usage (GPU box): bash tools/exp/npanel_sweep.sh <tag> -> isolated per-shape table + the N-panel walk on res5's shapes
TAG=${1:-r05}
cd $GRAFT_REPO_ROOT
python tools/bench_conv.py > gpurun_out/${TAG}_bench_conv.txt 2>&1
for np in 0 1 2 4 8 -1; do
  echo "== n_panel=$np" >> gpurun_out/${TAG}_npanel.txt
  BENCH_TUNE=n_panel=$np python tools/bench_conv.py res5 >> gpurun_out/${TAG}_npanel.txt 2>&1
done
for kb in 1024 3072; do
  echo "== n_panel=-1 n_panel_kb=$kb" >> gpurun_out/${TAG}_npanel.txt
  BENCH_TUNE=n_panel=-1,n_panel_kb=$kb python tools/bench_conv.py res5 >> gpurun_out/${TAG}_npanel.txt 2>&1
done
cat gpurun_out/${TAG}_npanel.txt | grep -v amdgpu.ids
