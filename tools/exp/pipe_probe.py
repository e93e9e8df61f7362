"""Developer probe: the software-pipelined 128x128 forward-form kernel (PIPE, one workgroup per CU;
mrcnn_set_tuning("pipe", 1)) against the standard instantiation: bit-identity on the same K
partition and time per shape, alone on the GPU (forward with the bottleneck epilogue and the
transposed-filter data gradient)."""
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
    ('res3 1x1 128->512', 2, 128, 101, 167, 512, 1, 1, 0),
    ('res4 3x3 256', 2, 256, 51, 84, 256, 3, 1, 1),
    ('res4 1x1 256->1024', 2, 256, 51, 84, 1024, 1, 1, 0),
    ('res4 1x1 1024->256', 2, 1024, 51, 84, 256, 1, 1, 0),
    ('res5 1x1 512->2048', 1024, 512, 7, 7, 2048, 1, 1, 0),
    ('res5 1x1 2048->512', 1024, 2048, 7, 7, 512, 1, 1, 0),
]


def main():
    lib = _lib.load()
    cfgs = [(0, 0), (0, 256), (1, 256), (0, 384), (1, 384), (0, 512), (1, 512), (1, 0)]
    print('%-22s' % 'shape' + ''.join('  pipe=%d k=%-4d' % c for c in cfgs) + '   (fwd | dgrad_wt us; * = differs from the non-PIPE run of the same k)')
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
        row, ref = '%-22s' % name, {}
        for pipe, kn in cfgs:
            _lib.check(lib.mrcnn_set_tuning(b'pipe', pipe), 'set_tuning')
            _lib.check(lib.mrcnn_set_tuning(b'big_split_k', kn), 'set_tuning')
            tf, tt = timeit(f, 20), timeit(t, 20)
            out = (y.clone(), gx.clone())
            mark = ''
            if pipe == 0:
                ref[kn] = out
            elif kn in ref and not (torch.equal(out[0], ref[kn][0]) and torch.equal(out[1], ref[kn][1])):
                mark = '*'
            row += ' %6.1f|%6.1f%1s' % (tf * 1e3, tt * 1e3, mark)
        print(row, flush=True)
    _lib.check(lib.mrcnn_set_tuning(b'big_split_k', 0), 'set_tuning')
    _lib.check(lib.mrcnn_set_tuning(b'pipe', 0), 'set_tuning')


if __name__ == '__main__':
    main()
