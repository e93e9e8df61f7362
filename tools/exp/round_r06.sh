# usage (GPU box): bash tools/exp/round_r06.sh <tag>  -> gpu tests, smoke, the round's profile collection (PMC passes
# BEFORE the bench lines that cite them: tools/collect_round_profiles.sh), probes and same-box A/Bs of round 6
TAG=${1:-r06b}
# (no shell trace: the A/B files are read as they are)
cd $GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu > gpurun_out/${TAG}_gputests.log 2>&1; grep -E "passed|failed" gpurun_out/${TAG}_gputests.log | tail -2
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${TAG}_smoke.log 2>&1; tail -2 gpurun_out/${TAG}_smoke.log
bash tools/collect_round_profiles.sh ${TAG} > gpurun_out/${TAG}_collect.log 2>&1
python tools/step_shapes.py > gpurun_out/${TAG}_step_shapes.txt 2>&1
python tools/exp/roi_proj_probe.py > gpurun_out/${TAG}_roi_proj_probe.txt 2>&1
python tools/exp/roi_align_probe.py > gpurun_out/${TAG}_roi_align_probe.txt 2>&1
python tools/exp/nms_probe.py > gpurun_out/${TAG}_nms_probe.txt 2>&1
python tools/bench_conv.py > gpurun_out/${TAG}_bench_conv.txt 2>&1
# W8 per shape (the RoI head's 1x1 layers and two backbone shapes it is NOT used for)
( for w in 0 1; do echo "== w8=$w"; for s in "res5 1x1" "res4 1x1 256" "res3 1x1 128"; do BENCH_TUNE=w8=$w python tools/bench_conv.py "$s" 2>/dev/null | grep -v "^shape\|^sum"; done; done ) > gpurun_out/${TAG}_w8_shapes.txt 2>&1
# same-box A/Bs of the round's step-level changes (headline region only, alternating arms, twice)
Q="--no-cpu-baseline --rotate-batches 0 --no-fg-capped --no-direct-head-forward --no-device-targets --pipeline-examples 0 --no-extra-workloads --no-fp32-mfma --repeats 3"
ab() { python bench.py $Q "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['repeats']['ms_per_step'], d['config'].get('loss'))"; }
( for rep in 1 2; do
    echo "== reference order (MRCNN_PROJECTED_POOLING=0, w8=0)"; MRCNN_PROJECTED_POOLING=0 ab --tune w8=0
    echo "== projected pooling, w8=0"; ab --tune w8=0
    echo "== projected pooling + W8 (default)"; ab
    echo "== default, --no-smi"; ab --no-smi
  done ) > gpurun_out/${TAG}_ab_round6.txt 2>&1
# inference kernel table (BASELINE configs[4])
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_infer_stats -o ${TAG}_infer -- python $GRAFT_REPO_ROOT/bench.py --workload infer --steps 4 --warmup 2 > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_infer_stats.log 2>&1
cd $GRAFT_REPO_ROOT
ls gpurun_out | grep ${TAG} | head -60
python tools/exp/train_curve.py > gpurun_out/${TAG}_train_curve.txt 2>&1
