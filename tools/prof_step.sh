# usage: bash tools/prof_step.sh <tag>   (on the GPU box, via gpurun)
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$1 -o $1 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-profile > $R/gpurun_out/$1.log 2>&1
ls $R/gpurun_out/$1
