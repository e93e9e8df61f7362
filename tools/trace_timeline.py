"""Developer tool: ordered per-kernel timeline of the LAST train step in a rocprofv3
--kernel-trace CSV (usage: python tools/trace_timeline.py <t_kernel_trace.csv> [n_steps_in_trace]).
Prints start offset, duration, queue, short kernel name and grid; then busy / idle totals."""
import csv, re, sys


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    # <TM, TN, MODE, MASKED, WPERM, SPLIT, W8, PW, K3>
    m = re.match(r'conv_gemm_kernel<(\d), (\d), (\d), (\w+), (\w+)(?:, (\w+))?(?:, (\w+))?(?:, (\w+))?'
                 r'(?:, (\w+))?>', name)
    if m:
        return 'gemm<%s%s,%s%s%s%s%s%s%s>' % (m.group(1), m.group(2), 'FDW'[int(m.group(3))],
                                             ',M' if m.group(4) == 'true' else '',
                                             ',P' if m.group(5) == 'true' else '',
                                             ',S' if m.group(6) == 'true' else '',
                                             ',W8' if m.group(7) == 'true' else '',
                                             ',PW' if m.group(8) == 'true' else '',
                                             ',K3' if m.group(9) == 'true' else '')
    name = re.sub(r'at::native::', 'at::', name)
    return name.split('(')[0][:60]


def main():
    rows = []
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Queue_Id'],
                         short(r['Kernel_Name']),
                         (int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])),
                          int(r['Grid_Size_Y']), int(r['Grid_Size_Z'])), r['VGPR_Count'],
                         r['LDS_Block_Size']))
    rows.sort()
    # steps are delimited by the extractor's max-pooling launch (one per step, right behind the
    # stem convolution)
    marks = [i for i, r in enumerate(rows) if r[3].startswith('maxpool_kernel')]
    lo, hi = marks[-2], marks[-1]
    step = rows[lo:hi]
    t0 = step[0][0]
    busy_end, idle = t0, 0
    for s, e, q, n, g, v, l in step:
        gap = s - busy_end
        if gap > 0:
            idle += gap
        print('%9.1f %8.1f  q%-2s %s%-46s %s' % ((s - t0) / 1e3, (e - s) / 1e3, q,
                                                  '  ' if gap <= 0 else '| ', n, g))
        busy_end = max(busy_end, e)
    print('step wall %.2f ms, idle (no kernel running) %.2f ms, kernels %d'
          % ((busy_end - t0) / 1e6, idle / 1e6, len(step)))
    agg = {}
    for s, e, q, n, g, v, l in step:
        a = agg.setdefault(n, [0, 0.])
        a[0] += 1
        a[1] += (e - s) / 1e6
    for n, (c, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
        print('%8.3f ms %4d  %s' % (ms, c, n))


if __name__ == '__main__':
    main()
