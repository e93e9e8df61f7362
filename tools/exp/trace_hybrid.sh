# usage (GPU box): bash tools/exp/trace_hybrid.sh <pct>
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/hyb_trace
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/hyb_trace -o t -- python $R/tools/exp/hybrid_check.py $1 > $R/gpurun_out/hyb_trace.log 2>&1
python - <<PY
import csv, glob
f = glob.glob('$R/gpurun_out/hyb_trace/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows = [r for r in rows if 'conv_gemm' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
t0 = int(rows[0]['Start_Timestamp'])
out = open('$R/gpurun_out/hyb_trace_summary.txt', 'w')
for r in rows[:60]:
    nm = r['Kernel_Name']
    kind = 'VALU' if 'false, false, true' in nm else ('MFMA128' if '<2, 2, 0' in nm else 'other')
    out.write('%-8s start %9.1f us  dur %8.1f us  grid %s lds %s q %s\n' % (
        kind, (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3,
        r.get('Grid_Size_X', r.get('Grid_Size', '?')), r.get('LDS_Block_Size', '?'), r.get('Queue_Id', '?')))
out.close()
PY
