# developer: kernel-trace durations of the ROIAlign kernels over one short bench run (in-step launches first, then the isolated repetitions)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/rt -o rt -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --rotate-batches 0 --no-fg-capped --no-device-targets --no-direct-head-forward --no-split-bf16 --pipeline-examples 0 > /tmp/rt.log 2>&1
python - <<'P'
import csv, glob
f = glob.glob('/tmp/rt/*kernel_trace.csv')[0]
rows = [r for r in csv.DictReader(open(f)) if 'roi_align' in r['Kernel_Name'] or 'roi_bwd' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
for kind in ('roi_align_fwd_kernel', 'roi_align_bwd_owner_kernel'):
    d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows if kind in r['Kernel_Name']]
    print(kind, len(d), ' '.join('%.0f' % v for v in d))
P
