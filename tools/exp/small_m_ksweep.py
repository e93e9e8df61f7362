"""Experiment: fixed cost vs per-K-slice cost of the 64x64-tile forward-form kernel on a res4-sized
problem (M = 2 x 51 x 84 rows): time(K) = a + b * K/32."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from chainer_mask_rcnn_amd import _lib
from chainer_mask_rcnn_amd.functions import conv as C
from chainer_mask_rcnn_amd.functions._layout import nhwc

dev = torch.device('cuda:0')


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


for (n, h, w) in ((2, 51, 84), (2, 101, 167)):
    for K in (256, 1024):
        for Cin in (64, 128, 256, 512, 1024, 2048, 4096):
            x = nhwc(torch.randn(n, Cin, h, w, device=dev))
            wt = nhwc(torch.randn(K, Cin, 1, 1, device=dev) * 0.05)
            d = C.make_desc(x.shape, wt.shape, 1, 0)
            us = timeit(lambda: C._fwd_raw(x, wt, d, None, None, None, False))
            fl = 2.0 * n * h * w * Cin * K
            print('M=%6d N=%5d K=%5d slices=%4d  %8.1f us  %6.1f TFLOP/s' % (n * h * w, K, Cin, Cin // 32, us, fl / us / 1e6))
