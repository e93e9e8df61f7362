set -x
cd $GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu > gpurun_out/r03f_gputests.log 2>&1; tail -3 gpurun_out/r03f_gputests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r03f_smoke.log 2>&1; tail -2 gpurun_out/r03f_smoke.log
python tools/exp/roi_align_probe.py > gpurun_out/r03f_roi_align_probe.txt 2>&1; PROBE_STEPS=26 python tools/exp/roi_align_probe.py > gpurun_out/r03f_roi_align_probe_26steps.txt 2>&1
bash tools/collect_round_profiles.sh r03f > gpurun_out/r03f_collect.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03f_roi_fetch -o roi -- PROBE_STEPS=26 python $GRAFT_REPO_ROOT/tools/exp/roi_align_probe.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03f_roi_write -o roi -- PROBE_STEPS=26 python $GRAFT_REPO_ROOT/tools/exp/roi_align_probe.py > /dev/null 2>&1
ls $GRAFT_REPO_ROOT/gpurun_out | grep r03f | head -40
