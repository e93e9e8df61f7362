# usage (GPU box): bash tools/exp/trace_step.sh <tag> [extra bench flags]  -> gpurun_out/<tag>_timeline.txt
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT
ONLY="--rotate-batches 0 --no-fg-capped --no-device-targets --no-direct-head-forward --no-fp32-mfma --pipeline-examples 0 --no-extra-workloads"
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/$TAG -o $TAG -- python $R/bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-profile $ONLY "$@" > $R/gpurun_out/$TAG.log 2>&1
python $R/tools/trace_timeline.py $(find $R/gpurun_out/$TAG -name "*kernel_trace.csv" | head -1) > $R/gpurun_out/${TAG}_timeline.txt 2>&1
rm -rf $R/gpurun_out/$TAG
tail -45 $R/gpurun_out/${TAG}_timeline.txt
