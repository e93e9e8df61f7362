cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d $R/gpurun_out/pmc_mfma -o p -- python $R/tools/bench_conv.py "res5 3x3" > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq -o p -- python $R/tools/bench_conv.py "res5 3x3" > /dev/null 2>&1
ls $R/gpurun_out/pmc_mfma $R/gpurun_out/pmc_sq
