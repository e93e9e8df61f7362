"""Developer microbenchmark: the three passes of a 3x3 / pad 1 convolution over many small maps,
Winograd F(4x4,3x3) route vs the implicit-GEMM route.  usage: python tools/bench_winograd.py [N C K H]"""
import sys
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chainer_mask_rcnn_amd import _lib
from chainer_mask_rcnn_amd.functions import conv as C
from chainer_mask_rcnn_amd.functions._layout import nhwc, empty_nhwc

N, Cc, K, H = (int(a) for a in sys.argv[1:5]) if len(sys.argv) >= 5 else (1024, 512, 512, 7)
dev = torch.device('cuda:0')
x = nhwc(torch.randn(N, Cc, H, H, device=dev).relu_())
w = nhwc(torch.randn(K, Cc, 3, 3, device=dev) * 0.02)
g = nhwc(torch.randn(N, K, H, H, device=dev))
sc = torch.rand(K, device=dev) + 0.5
sh = torch.randn(K, device=dev)
m = nhwc(torch.randn(N, Cc, H, H, device=dev))
d = C.make_desc(x.shape, w.shape, 1, 1)
gW = torch.empty_like(w)
_, v = C.wino_fwd(x, w, d, sc, sh, True, keep_v=True)
wT = torch.empty(w.numel(), device=dev)
ws = _lib.workspace(_lib.load().mrcnn_conv2d_wgrad_workspace_bytes(C.ctx_desc(d)), dev, 'wgrad')


def direct_wgrad():
    _lib.call('mrcnn_conv2d_wgrad_ex', C.ctx_desc(d), _lib.ptr(x), _lib.ptr(g), _lib.ptr(gW),
              _lib.ptr(ws), None, None, None, _lib.stream_ptr())


def direct_dgrad():
    C._flip_transpose(w, d, None, wT)
    C._dgrad_raw(d, g, w, None, None, out_mask_y=m, out_scale=sc[:Cc] if Cc <= K else None, wT=wT)


cases = [
    ('fwd   direct', lambda: C._fwd_raw(x, w, d, sc, sh, None, True)),
    ('fwd   wino  ', lambda: C.wino_fwd(x, w, d, sc, sh, True, keep_v=True)),
    ('dgrad direct', direct_dgrad),
    ('dgrad wino  ', lambda: C.wino_dgrad(d, g, w, out_scale=sc[:Cc] if Cc <= K else None, out_mask_y=m)),
    ('wgrad direct', direct_wgrad),
    ('wgrad wino v', lambda: C.wino_wgrad_into(d, None, v, g, gW)),
    ('wgrad wino x', lambda: C.wino_wgrad_into(d, x, None, g, gW)),
]
flops = 2.0 * N * H * H * Cc * K * 9
for name, fn in cases:
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print('%s  %.3f ms   %.1f direct-equivalent TFLOP/s' % (name, ms, flops / ms / 1e9))
