# usage (GPU box): bash tools/exp/pmc_split.sh "<bench_conv filter>"  -> PMC passes of the split-operand kernels
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT
export BENCH_TUNE=${BENCH_TUNE:-split_bf16=3}
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU" \
           "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_split_$i -o p -- python $R/tools/bench_conv.py "$1" > /tmp/pmc_$i.log 2>&1 || tail -3 /tmp/pmc_$i.log
  i=$((i+1))
done
python - <<PY
import csv, glob, collections
for d in sorted(glob.glob('$R/gpurun_out/pmc_split_*')):
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'][:70]
            if 'conv_gemm' not in k: continue
            acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        for k, v in acc.items():
            print(k); [print('   %-34s %.4g' % (c, x)) for c, x in sorted(v.items())]
PY
