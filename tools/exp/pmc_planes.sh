# usage (GPU box): bash tools/exp/pmc_planes.sh "<planes_probe filter>" "<modes>"  -> memory-side PMC passes of one GEMM shape
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT
export PROBE_MODES=${2:-0}
i=0
for set in "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM" \
           "TCC_EA0_RDREQ_32B_sum TCC_EA0_RD_UNCACHED_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_BUBBLE_sum"; do
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_planes_$i -o p -- python $R/tools/exp/planes_probe.py "$1" > /tmp/pmc_$i.log 2>&1 || tail -3 /tmp/pmc_$i.log
  i=$((i+1))
done
python - <<PY
import csv, glob, collections
for d in sorted(glob.glob('$R/gpurun_out/pmc_planes_*')):
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'][:90]
            if 'conv_gemm' not in k: continue
            acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[k].add(r['Dispatch_Id'])
        for k, v in acc.items():
            print(k, 'dispatches', len(n[k])); [print('   %-34s %.4g per dispatch' % (c, x / len(n[k]))) for c, x in sorted(v.items())]
PY
