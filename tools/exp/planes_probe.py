"""Developer probe: forward-form split-operand GEMMs with pre-split operand planes
(include/mrcnn_hip.h "operand planes") against the in-kernel split — bit equality of the
results and per-shape TFLOP/s of (no planes | filter planes | both operands)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from chainer_mask_rcnn_amd import _lib
from chainer_mask_rcnn_amd.functions.conv import make_desc, ctx_desc, split_ws
from chainer_mask_rcnn_amd.functions._layout import empty_nhwc

dev = torch.device('cuda:0')
SHAPES = [
    ('res4 3x3 256', 2, 256, 51, 84, 256, 3, 1, 1),
    ('res4 1x1 256->1024', 2, 256, 51, 84, 1024, 1, 1, 0),
    ('res4 1x1 1024->256', 2, 1024, 51, 84, 256, 1, 1, 0),
    ('rpn 3x3 1024', 2, 1024, 51, 84, 1024, 3, 1, 1),
    ('res5a 1x1s2 1024->512', 1024, 1024, 14, 14, 512, 1, 2, 0),
    ('res5a 1x1 1024->2048 (binned)', 1024, 1024, 7, 7, 2048, 1, 1, 0),
    ('res5 3x3 512', 1024, 512, 7, 7, 512, 3, 1, 1),
    ('res5 1x1 512->2048', 1024, 512, 7, 7, 2048, 1, 1, 0),
    ('res5 1x1 2048->512', 1024, 2048, 7, 7, 512, 1, 1, 0),
    ('wino-like 1x1 512->512 M=65536', 1024, 512, 8, 8, 512, 1, 1, 0),
]


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def planes_of(t2d_rows, row_len, src):
    pl = torch.empty((t2d_rows * row_len * 6,), dtype=torch.uint8, device=dev)
    _lib.call('mrcnn_split_planes', _lib.ptr(src), _lib.ptr(pl), t2d_rows, row_len, _lib.stream_ptr())
    return pl


MODES = [int(m) for m in os.environ.get('PROBE_MODES', '0,1,2').split(',')]


def main():
    only = sys.argv[1] if len(sys.argv) > 1 else None
    lib = _lib.load()
    _lib.set_tuning('split_bf16', 3)
    print('%-34s %14s %14s %14s %8s' % ('shape', 'in-kernel', 'filter planes', 'both planes', 'equal'))
    for name, N, C, H, W, K, k, s, p in SHAPES:
        if only and only not in name:
            continue
        PAD = int(os.environ.get('PROBE_PITCH', 0))     # variant builds with -DMRCNN_DBG_PITCH read padded rows
        x = torch.randn((N, H, W, C + PAD), device=dev)[..., :C].permute(0, 3, 1, 2)
        w = (torch.randn((K, k, k, C + PAD), device=dev) * 0.05)[..., :C].permute(0, 3, 1, 2)
        if PAD:
            x = torch.randn((N * H * W * (C + PAD),), device=dev)[:N * H * W * C].view(N, H, W, C).permute(0, 3, 1, 2)
            w = (torch.randn((K * k * k * (C + PAD),), device=dev) * 0.05)[:K * k * k * C].view(K, k, k, C).permute(0, 3, 1, 2)
        d = make_desc(x.shape, w.shape, s, p)
        esc = torch.rand((K,), device=dev) + 0.5
        esh = torch.randn((K,), device=dev)
        flop = 2.0 * d.N * d.P * d.Q * K * C * k * k
        sw, sp = _lib.ptr(split_ws(dev)), _lib.stream_ptr()
        x_pl = planes_of(N * H * W, C, x)
        w_pl = planes_of(K, k * k * C, w)
        outs, times = [], []
        for mode in MODES:
            y = empty_nhwc((d.N, d.K, d.P, d.Q), dev)
            y_pl = torch.zeros((d.N * d.P * d.Q * K * 6,), dtype=torch.uint8, device=dev)
            pl = _lib.Planes(_lib.ptr(x_pl) if mode == 2 else None,
                             _lib.ptr(w_pl) if mode >= 1 else None, _lib.ptr(y_pl))
            f = lambda: _lib.call('mrcnn_conv2d_fwd_pl', ctx_desc(d), _lib.ptr(x), _lib.ptr(w), None,
                                  _lib.ptr(esc), _lib.ptr(esh), None, _lib.ptr(y), 2 | 8, sw,
                                  ctypes.byref(pl), sp)
            times.append(timeit(f))
            ref_pl = planes_of(d.N * d.P * d.Q, K, y)
            outs.append((y.clone(), bool(torch.equal(ref_pl, y_pl))))
        eq = all(torch.equal(outs[0][0], o[0]) for o in outs[1:]) and all(o[1] for o in outs)
        print('%-34s %s %8s' % (name, ' '.join('%6.1f|%6.3f' % (flop / t / 1e9, t) for t in times), eq))


if __name__ == '__main__':
    main()
