"""MFMA-pipe utilisation of the conv GEMM kernels from the PMC passes of tools/pmc_conv.sh
(one `tools/bench_conv.py "<shape>"` run per shape: forward, dgrad, wgrad and transposed-filter
dgrad of that layer), written to profiles/<tag>_pmc_mfma.json.

utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs);
sustained clock = (GRBM_GUI_ACTIVE / 8) / kernel duration."""
import collections, csv, glob, json, os, sys
tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
g = os.path.join(root, 'gpurun_out')
result = collections.OrderedDict()
for shape_file in sorted(glob.glob(os.path.join(g, 'pmc_*_shape.txt'))):
    idx = os.path.basename(shape_file).split('_')[1]
    shape = open(shape_file).read().strip()
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in ('pmc_%s_mfma' % idx, 'pmc_%s_sq' % idx):
        files = glob.glob(os.path.join(g, d, '**', '*counter_collection.csv'), recursive=True)
        if not files:
            continue
        for r in csv.DictReader(open(files[0])):
            if 'conv_gemm_kernel' in r['Kernel_Name']:
                name = 'conv_gemm_kernel' + r['Kernel_Name'].split('conv_gemm_kernel')[1].split('(')[0]
                acc[name][r['Counter_Name']].append(float(r['Counter_Value']))
    dur = collections.defaultdict(list)
    files = glob.glob(os.path.join(g, 'pmc_%s_mfma' % idx, '**', '*kernel_trace.csv'), recursive=True)
    if files:
        for r in csv.DictReader(open(files[0])):
            if 'conv_gemm_kernel' in r['Kernel_Name']:
                name = 'conv_gemm_kernel' + r['Kernel_Name'].split('conv_gemm_kernel')[1].split('(')[0]
                dur[name].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    out = {}
    for k, v in acc.items():
        m = {c: sum(x) / len(x) for c, x in v.items()}
        if 'GRBM_GUI_ACTIVE' not in m or not dur[k]:
            continue
        cyc = m['GRBM_GUI_ACTIVE'] / 8.
        ns = sum(dur[k]) / len(dur[k])
        out[k] = dict(launches=len(v['GRBM_GUI_ACTIVE']), avg_duration_us=round(ns / 1e3, 1),
                      sustained_clock_ghz=round(cyc / ns, 3),
                      mfma_instructions=m.get('SQ_INSTS_MFMA'), valu_instructions=m.get('SQ_INSTS_VALU'),
                      mfma_pipe_utilisation=round(m['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024. * cyc), 4),
                      wave_cycles_waiting_any_frac=round(m['SQ_WAIT_ANY'] / m['SQ_WAVE_CYCLES'], 4)
                      if 'SQ_WAVE_CYCLES' in m else None,
                      wave_cycles_waiting_lds_frac=round(m['SQ_WAIT_INST_LDS'] / m['SQ_WAVE_CYCLES'], 4)
                      if 'SQ_WAVE_CYCLES' in m else None)
    result[shape] = out
json.dump(dict(workload='tools/bench_conv.py "<shape>" (forward, dgrad, wgrad, transposed-filter dgrad of '
                        'one layer shape at the BASELINE configs[1] size) under rocprofv3 --pmc, two '
                        'passes per shape (tools/pmc_conv.sh)', shapes=result),
          open(os.path.join(root, 'profiles', '%s_pmc_mfma.json' % tag), 'w'), indent=1)
print(json.dumps(result, indent=1)[:3000])
