# usage (GPU box): bash tools/pmc_conv.sh   -> gpurun_out/pmc_<n>_{mfma,sq}/ for each shape below
# Two PMC passes per shape of tools/bench_conv.py (separate runs, --kernel-trace only with --pmc).
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT
i=0
for shape in "res5 3x3" "res5 1x1 2048->512" "res5 1x1 512->2048" "rpn 3x3" "res4 3x3" "res4 1x1 1024->256" "res4 1x1 256->1024" "res3 3x3"; do
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${i}_mfma -o p -- python $R/tools/bench_conv.py "$shape" > /dev/null 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${i}_sq -o p -- python $R/tools/bench_conv.py "$shape" > /dev/null 2>&1
  echo "$shape" > $R/gpurun_out/pmc_${i}_shape.txt
  i=$((i+1))
done
ls $R/gpurun_out | grep pmc_
