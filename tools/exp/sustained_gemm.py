"""Experiment: does the forward-form GEMM's rate depend on how long the GPU has been under load
(power management)?  Times the res5 1x1 2048->512 convolution over 10 / 100 / 1000 / 3000 launches."""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from chainer_mask_rcnn_amd.functions import conv as C
from chainer_mask_rcnn_amd.functions._layout import nhwc

dev = torch.device('cuda:0')
for (cin, cout) in ((2048, 512), (512, 2048)):
    x = nhwc(torch.randn(1024, cin, 7, 7, device=dev))
    w = nhwc(torch.randn(cout, cin, 1, 1, device=dev) * 0.02)
    d = C.make_desc(x.shape, w.shape, 1, 0)
    fl = 2.0 * 1024 * 49 * cin * cout
    for iters in (10, 100, 1000, 3000):
        torch.cuda.synchronize()
        time.sleep(0.5)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            C._fwd_raw(x, w, d, None, None, None, False)
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / iters
        print('%d->%d  %5d launches  %.3f ms  %.1f TFLOP/s' % (cin, cout, iters, ms, fl / ms / 1e9))
