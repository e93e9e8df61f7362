# usage (GPU box): bash tools/exp/round_r04.sh <tag>  -> gpu tests, smoke, ROIAlign probes and the round's profile collection
TAG=${1:-r04a}
set -x
cd $GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu > gpurun_out/${TAG}_gputests.log 2>&1; tail -3 gpurun_out/${TAG}_gputests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${TAG}_smoke.log 2>&1; tail -2 gpurun_out/${TAG}_smoke.log
python tools/exp/roi_align_probe.py > gpurun_out/${TAG}_roi_align_probe.txt 2>&1

./tools/exp/ldmfma_probe > gpurun_out/${TAG}_ldmfma_probe.txt 2>&1
bash tools/collect_round_profiles.sh ${TAG} > gpurun_out/${TAG}_collect.log 2>&1
ls gpurun_out | grep ${TAG} | head -40
