cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r01f_infer -o r01f_infer -- python $R/bench.py --workload infer --steps 3 --warmup 1 > $R/gpurun_out/r01f_infer.log 2>&1
