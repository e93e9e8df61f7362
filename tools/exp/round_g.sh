# final refresh after the kernel-timer change: default bench line, every-kind line, rocprofv3 kernel stats of the default command
R=$GRAFT_REPO_ROOT
ONLY="--rotate-batches 0 --no-fg-capped --no-device-targets --no-direct-head-forward --no-split-bf16 --pipeline-examples 0"
python $R/bench.py > $R/gpurun_out/r03g_bench.json 2> $R/gpurun_out/r03g_bench.err
python $R/bench.py --profile-all --no-cpu-baseline $ONLY > $R/gpurun_out/r03g_bench_allkinds.json 2>> $R/gpurun_out/r03g_bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03g_stats -o r03g -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/r03g_stats.log 2>&1
ls $R/gpurun_out | grep r03g
