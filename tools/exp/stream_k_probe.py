"""Developer probe: small-M forward-form GEMMs (the batch-2 backbone: res3 / res4) as a stream-K launch of
the 128x128 kernel (mrcnn_set_tuning("stream_k", workgroups): every workgroup a contiguous range of
(tile, K slice) units) against the shipped 64x64-tile policy: per shape the time of the forward and of
the transposed-filter data gradient and the largest difference of the results (summation order only)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from chainer_mask_rcnn_amd import _lib
from chainer_mask_rcnn_amd.functions.conv import make_desc, ctx_desc, split_ws
from chainer_mask_rcnn_amd.functions._layout import empty_nhwc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_conv import timeit

dev = torch.device('cuda:0')
SHAPES = [
    ('res3 3x3 128', 2, 128, 101, 167, 128, 3, 1, 1),
    ('res3 1x1 512->128', 2, 512, 101, 167, 128, 1, 1, 0),
    ('res4 3x3 256', 2, 256, 51, 84, 256, 3, 1, 1),
    ('res4 1x1 256->1024', 2, 256, 51, 84, 1024, 1, 1, 0),
    ('res4 1x1 1024->256', 2, 1024, 51, 84, 256, 1, 1, 0),
    ('res2 3x3 64', 2, 64, 201, 334, 64, 3, 1, 1),
]


def main():
    lib = _lib.load()
    knobs = [int(v) for v in (sys.argv[1:] or ['0', '256', '384', '448', '512', '768'])]
    print('%-22s' % 'shape' + ''.join('  k=%-4d fwd | dgrad_wt us' % k for k in knobs))
    for name, N, C, H, W, K, k, s, p in SHAPES:
        x = torch.randn((N, H, W, C), device=dev).permute(0, 3, 1, 2)
        w = (torch.randn((K, k, k, C), device=dev) * 0.05).permute(0, 3, 1, 2)
        d = make_desc(x.shape, w.shape, s, p)
        y = empty_nhwc((d.N, d.K, d.P, d.Q), dev)
        gy = torch.randn((d.N, d.P, d.Q, d.K), device=dev).permute(0, 3, 1, 2)
        gx = empty_nhwc((N, C, H, W), dev)
        sc = torch.rand((K,), device=dev) + 0.5
        sh = torch.randn((K,), device=dev)
        res = torch.randn((d.N, d.P, d.Q, d.K), device=dev).permute(0, 3, 1, 2)
        xm = torch.randn((N, H, W, C), device=dev).permute(0, 3, 1, 2)
        sc2 = torch.rand((C,), device=dev) + 0.5
        wT = torch.empty((C * k * k * K,), device=dev)
        sp, sw = _lib.stream_ptr(), _lib.ptr(split_ws(dev))
        _lib.call('mrcnn_filter_flip_transpose', _lib.ptr(w), _lib.ptr(wT), K, k, k, C, None, sp)
        f = lambda: _lib.call('mrcnn_conv2d_fwd', ctx_desc(d), _lib.ptr(x), _lib.ptr(w), None, _lib.ptr(sc),
                              _lib.ptr(sh), _lib.ptr(res), _lib.ptr(y), 2 | 4 | 8, sw, sp)
        t = lambda: _lib.call('mrcnn_conv2d_dgrad_wt', ctx_desc(d), _lib.ptr(gy), _lib.ptr(wT), _lib.ptr(gx), 0,
                              None, None, None, None, _lib.ptr(xm), _lib.ptr(sc2), sw, sp)
        row, ref = '%-22s' % name, None
        for kn in knobs:
            _lib.check(lib.mrcnn_set_tuning(b'stream_k', kn), 'set_tuning')
            tf, tt = timeit(f, 20), timeit(t, 20)
            out = (y.clone(), gx.clone())
            if ref is None:
                ref = out
            dy = float((out[0] - ref[0]).abs().max() / ref[0].abs().max())
            dg = float((out[1] - ref[1]).abs().max() / ref[1].abs().max())
            row += '  %6.1f | %6.1f (%.0e)' % (tf * 1e3, tt * 1e3, max(dy, dg))
        print(row, flush=True)
    _lib.check(lib.mrcnn_set_tuning(b'stream_k', 0), 'set_tuning')


if __name__ == '__main__':
    main()
