"""Copy the judged summaries of a gpurun profile collection into profiles/ (tracked)."""
import collections, csv, json, os, shutil, sys
tag = sys.argv[1]
pmc_only = '--pmc-only' in sys.argv     # (on the GPU box, before the bench lines that cite the summary)
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
g, p = os.path.join(root, 'gpurun_out'), os.path.join(root, 'profiles')
os.makedirs(p, exist_ok=True)
if not pmc_only:
    shutil.copy(os.path.join(g, '%s_stats' % tag, '%s_kernel_stats.csv' % tag),
                os.path.join(p, '%s_kernel_stats.csv' % tag))
for name in (() if pmc_only else ('bench', 'bench_allkinds', 'bench_infer', 'bench_dp1', 'bench_r101', 'bench_fp32_allkinds',
             'bench_fp32_r101')):
    src = os.path.join(g, '%s_%s.json' % (tag, name))
    if os.path.exists(src):
        lines = [l for l in open(src).read().splitlines() if l.startswith('{')]
        open(os.path.join(p, '%s_%s.json' % (tag, name)), 'w').write(lines[-1] + '\n')


def agg(path, cname):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == cname:
            acc[r['Kernel_Name']][0] += 1
            acc[r['Kernel_Name']][1] += float(r['Counter_Value'])
    return acc


src = os.path.join(g, '%s_fp32_stats' % tag, '%s_fp32_kernel_stats.csv' % tag)
if os.path.exists(src) and not pmc_only:
    shutil.copy(src, os.path.join(p, '%s_fp32_kernel_stats.csv' % tag))
f = agg(os.path.join(g, '%s_fetch' % tag, '%s_counter_collection.csv' % tag), 'FETCH_SIZE')
w = agg(os.path.join(g, '%s_write' % tag, '%s_counter_collection.csv' % tag), 'WRITE_SIZE')
out = {k: {'launches': f[k][0], 'FETCH_SIZE_KB_per_launch': f[k][1] / f[k][0],
           'WRITE_SIZE_KB_per_launch': (w[k][1] / w[k][0]) if k in w else None} for k in f}
json.dump(out, open(os.path.join(p, '%s_pmc_fetch_write.json' % tag), 'w'), indent=1)
print('profiles/%s_* written (%d kernels with PMC)' % (tag, len(out)))
