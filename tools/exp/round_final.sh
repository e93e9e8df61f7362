set -x
cd $GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu > gpurun_out/r03h_gputests.log 2>&1; tail -3 gpurun_out/r03h_gputests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r03h_smoke.log 2>&1; tail -2 gpurun_out/r03h_smoke.log
python tools/exp/roi_align_probe.py > gpurun_out/r03h_roi_align_probe.txt 2>&1; PROBE_STEPS=26 python tools/exp/roi_align_probe.py > gpurun_out/r03h_roi_align_probe_26steps.txt 2>&1
bash tools/collect_round_profiles.sh r03h > gpurun_out/r03h_collect.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03h_roi_fetch -o roi -- true
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03h_roi_write -o roi -- true
ls $GRAFT_REPO_ROOT/gpurun_out | grep r03h | head -40
