"""Developer tool: every convolution-GEMM launch of one train step, timed in place (HIP events
around each C-ABI call), grouped by (entry point, shape).  Shows which layer shapes the step
really spends its time on and at what TFLOP/s."""
import os, sys, collections
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from chainer_mask_rcnn_amd import _lib

GEMMS = ('mrcnn_conv2d_fwd', 'mrcnn_conv2d_dgrad', 'mrcnn_conv2d_dgrad_ex', 'mrcnn_conv2d_dgrad_wt',
         'mrcnn_conv2d_wgrad', 'mrcnn_conv2d_wgrad_ex')


def main():
    dev = torch.device('cuda:0')
    import random
    random.seed(0); np.random.seed(0); torch.manual_seed(0)
    rng = np.random.RandomState(0)
    imgs, bboxes, labels, masks, scales = bench.synthetic_batch(rng, 2, 800, 1333)
    model, chain, opt, sync = bench.build_trainer(50, dev, 1, 2)
    imgs_d = torch.tensor(imgs, device=dev).contiguous(memory_format=torch.channels_last)
    for _ in range(3):
        opt.update(chain, imgs_d, bboxes, labels, masks, scales)
    torch.cuda.synchronize()
    records = []
    orig = _lib.call

    def traced(name, *args):
        if name not in GEMMS:
            return orig(name, *args)
        d = args[0]._obj
        key = (name.replace('mrcnn_conv2d_', ''), d.N, d.H, d.W, d.C, d.K, d.R, d.stride)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        orig(name, *args)
        b.record()
        records.append((key, a, b))

    _lib.call = traced
    import chainer_mask_rcnn_amd.functions.conv as C
    n = 3
    for _ in range(n):
        opt.update(chain, imgs_d, bboxes, labels, masks, scales)
    torch.cuda.synchronize()
    _lib.call = orig
    agg = collections.OrderedDict()
    for key, a, b in records:
        t = a.elapsed_time(b)
        c, s = agg.get(key, (0, 0.))
        agg[key] = (c + 1, s + t)
    rows = []
    for key, (c, s) in agg.items():
        name, N, H, W, Cc, K, R, st = key
        P = (H + 2 * (R // 2) - R) // st + 1
        Q = (W + 2 * (R // 2) - R) // st + 1
        flop = 2.0 * N * P * Q * K * Cc * R * R
        rows.append((s / n, c / n, name, key[1:], flop * c / s / 1e9))
    rows.sort(reverse=True)
    tot = sum(r[0] for r in rows)
    print('total GEMM ms/step (event-bracketed, includes launch gaps): %.2f' % tot)
    print('%8s %5s  %-10s %-34s %7s' % ('ms/step', 'calls', 'entry', '(N,H,W,C,K,R,stride)', 'TFLOP/s'))
    for ms, c, name, shape, tf in rows:
        print('%8.3f %5.1f  %-10s %-34s %7.1f' % (ms, c, name, shape, tf))


if __name__ == '__main__':
    main()
