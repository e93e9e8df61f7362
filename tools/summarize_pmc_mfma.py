"""MFMA-pipe utilisation of the conv GEMM kernels from the PMC passes of tools/pmc_conv.sh
(res5 3x3, 1024 RoIs: `tools/bench_conv.py "res5 3x3"`), written to profiles/<tag>_pmc_mfma.json.

utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs);
sustained clock = (GRBM_GUI_ACTIVE / 8) / kernel duration."""
import collections, csv, glob, json, os, sys
tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
g = os.path.join(root, 'gpurun_out')
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ('pmc_mfma', 'pmc_sq'):
    for r in csv.DictReader(open(glob.glob(os.path.join(g, d, '*counter_collection.csv'))[0])):
        if 'conv_gemm_kernel<2, 2' in r['Kernel_Name']:
            name = r['Kernel_Name'].split('conv_gemm_kernel')[1].split('(')[0]
            acc['conv_gemm_kernel' + name][r['Counter_Name']].append(float(r['Counter_Value']))
dur = collections.defaultdict(list)
for r in csv.DictReader(open(glob.glob(os.path.join(g, 'pmc_mfma', '*kernel_trace.csv'))[0])):
    if 'conv_gemm_kernel<2, 2' in r['Kernel_Name']:
        name = 'conv_gemm_kernel' + r['Kernel_Name'].split('conv_gemm_kernel')[1].split('(')[0]
        dur[name].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
out = {}
for k, v in acc.items():
    m = {c: sum(x) / len(x) for c, x in v.items()}
    cyc = m['GRBM_GUI_ACTIVE'] / 8.
    ns = sum(dur[k]) / len(dur[k])
    out[k] = dict(launches=len(v['GRBM_GUI_ACTIVE']), avg_duration_us=round(ns / 1e3, 1),
                  sustained_clock_ghz=round(cyc / ns, 3),
                  mfma_instructions=m['SQ_INSTS_MFMA'], valu_instructions=m['SQ_INSTS_VALU'],
                  mfma_pipe_utilisation=round(m['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024. * cyc), 4),
                  wave_cycles_waiting_any_frac=round(m['SQ_WAIT_ANY'] / m['SQ_WAVE_CYCLES'], 4),
                  wave_cycles_waiting_lds_frac=round(m['SQ_WAIT_INST_LDS'] / m['SQ_WAVE_CYCLES'], 4))
json.dump(dict(workload='res5 3x3 512->512 on 1024 RoIs (7x7), tools/bench_conv.py "res5 3x3" '
                        'under rocprofv3 --pmc (two passes)', kernels=out),
          open(os.path.join(root, 'profiles', '%s_pmc_mfma.json' % tag), 'w'), indent=1)
print(json.dumps(out, indent=1))
