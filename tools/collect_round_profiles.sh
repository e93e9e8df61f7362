# usage (on the GPU box via gpurun): bash tools/collect_round_profiles.sh <tag>
# Produces under gpurun_out/<tag>_*: FIRST the FETCH_SIZE / WRITE_SIZE PMC passes (resident-batch segment
# only) and their summary profiles/<tag>_pmc_fetch_write.json on the box — so that every bench line
# written afterwards cites THIS collection's traffic figure —, then the default bench JSON, the
# every-kind line, rocprofv3 kernel stats (csv) of the default command, and the other legs.
TAG=$1
R=$GRAFT_REPO_ROOT
ONLY="--rotate-batches 0 --no-fg-capped --no-device-targets --no-direct-head-forward --no-fp32-mfma --pipeline-examples 0 --no-extra-workloads"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_fetch -o ${TAG} -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile $ONLY > $R/gpurun_out/${TAG}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_write -o ${TAG} -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile $ONLY > $R/gpurun_out/${TAG}_write.log 2>&1
python $R/tools/summarize_profiles.py ${TAG} --pmc-only
cd $R
python $R/bench.py > $R/gpurun_out/${TAG}_bench.json 2> $R/gpurun_out/${TAG}_bench.err
python $R/bench.py --profile-all --no-cpu-baseline $ONLY > $R/gpurun_out/${TAG}_bench_allkinds.json 2>> $R/gpurun_out/${TAG}_bench.err
python $R/bench.py --workload infer --steps 3 --warmup 1 > $R/gpurun_out/${TAG}_bench_infer.json 2>> $R/gpurun_out/${TAG}_bench.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats -o ${TAG} -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/${TAG}_stats.log 2>&1
cd $R
ls $R/gpurun_out | grep ${TAG}
python $R/bench.py --force-dp --no-cpu-baseline $ONLY > $R/gpurun_out/${TAG}_bench_dp1.json 2>> $R/gpurun_out/${TAG}_bench.err
python $R/bench.py --layers 101 --no-cpu-baseline $ONLY > $R/gpurun_out/${TAG}_bench_r101.json 2>> $R/gpurun_out/${TAG}_bench.err
# the fp32-MFMA kernels (the default arithmetic up to round 3): bench line with every GEMM kind timed + kernel stats
python $R/bench.py --profile-all --no-cpu-baseline $ONLY --tune split_bf16=0 > $R/gpurun_out/${TAG}_bench_fp32_allkinds.json 2>> $R/gpurun_out/${TAG}_bench.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_fp32_stats -o ${TAG}_fp32 -- python $R/bench.py --no-cpu-baseline $ONLY --tune split_bf16=0 > $R/gpurun_out/${TAG}_fp32_stats.log 2>&1
cd $R
python $R/bench.py --layers 101 --no-cpu-baseline $ONLY --tune split_bf16=0 > $R/gpurun_out/${TAG}_bench_fp32_r101.json 2>> $R/gpurun_out/${TAG}_bench.err
