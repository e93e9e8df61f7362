# usage (GPU box): bash tools/exp/round_r05.sh <tag>  -> gpu tests, smoke, probes and the round's profile collection
TAG=${1:-r05a}
set -x
cd $GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu > gpurun_out/${TAG}_gputests.log 2>&1; tail -3 gpurun_out/${TAG}_gputests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${TAG}_smoke.log 2>&1; tail -2 gpurun_out/${TAG}_smoke.log
python tools/exp/roi_align_probe.py > gpurun_out/${TAG}_roi_align_probe.txt 2>&1
python tools/exp/nms_probe.py > gpurun_out/${TAG}_nms_probe.txt 2>&1
python tools/bench_conv.py > gpurun_out/${TAG}_bench_conv.txt 2>&1
bash tools/collect_round_profiles.sh ${TAG} > gpurun_out/${TAG}_collect.log 2>&1
# inference kernel table (BASELINE configs[4])
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_infer_stats -o ${TAG}_infer -- python $GRAFT_REPO_ROOT/bench.py --workload infer --steps 4 --warmup 2 > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_infer_stats.log 2>&1
cd $GRAFT_REPO_ROOT
ls gpurun_out | grep ${TAG} | head -60
