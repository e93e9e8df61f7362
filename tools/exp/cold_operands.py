"""Experiment: forward-form GEMM rate with the SAME operand buffers every launch (Infinity Cache /
TLB warm) vs rotating over R distinct input and output buffers (cold, as inside a train step)."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from chainer_mask_rcnn_amd import _lib
from chainer_mask_rcnn_amd.functions import conv as C
from chainer_mask_rcnn_amd.functions._layout import nhwc, empty_nhwc
from chainer_mask_rcnn_amd._lib import EPI_RELU

dev = torch.device('cuda:0')
for (cin, cout) in ((2048, 512), (512, 2048)):
    fl = 2.0 * 1024 * 49 * cin * cout
    w = nhwc(torch.randn(cout, cin, 1, 1, device=dev) * 0.02)
    for R in (1, 2, 4, 8, 16):
        xs = [nhwc(torch.randn(1024, cin, 7, 7, device=dev)) for _ in range(R)]
        ys = [empty_nhwc((1024, cout, 7, 7), dev) for _ in range(R)]
        d = C.make_desc(xs[0].shape, w.shape, 1, 0)
        ws = C.split_ws(dev)

        def run(i):
            _lib.call('mrcnn_conv2d_fwd', C.ctx_desc(d), _lib.ptr(xs[i % R]), _lib.ptr(w), None, None,
                      None, None, _lib.ptr(ys[i % R]), 0, _lib.ptr(ws), _lib.stream_ptr())
        for i in range(20):
            run(i)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        iters = 200
        a.record()
        for i in range(iters):
            run(i)
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / iters
        print('%d->%d  %2d buffer sets (%5.0f MB in flight)  %.3f ms  %.1f TFLOP/s' % (
            cin, cout, R, R * 1024 * 49 * (cin + cout) * 4 / 1e6, ms, fl / ms / 1e9))
        del xs, ys
