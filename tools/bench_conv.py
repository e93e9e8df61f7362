"""Developer tool: per-shape TFLOP/s of the implicit-GEMM conv family on the C2 shapes."""
import sys, os, ctypes
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chainer_mask_rcnn_amd import _lib
from chainer_mask_rcnn_amd.functions.conv import make_desc, ctx_desc, split_ws
from chainer_mask_rcnn_amd.functions._layout import empty_nhwc

dev = torch.device('cuda:0')
SHAPES = [
    # name, N, C, H, W, K, k, stride, pad
    ('stem-like res2 1x1 64->256', 2, 64, 201, 334, 256, 1, 1, 0),
    ('res2 3x3 64', 2, 64, 201, 334, 64, 3, 1, 1),
    ('res3a 1x1s2 256->128', 2, 256, 201, 334, 128, 1, 2, 0),
    ('res3 3x3 128', 2, 128, 101, 167, 128, 3, 1, 1),
    ('res3 1x1 128->512', 2, 128, 101, 167, 512, 1, 1, 0),
    ('res3 1x1 512->128', 2, 512, 101, 167, 128, 1, 1, 0),
    ('res4a 1x1s2 512->256', 2, 512, 101, 167, 256, 1, 2, 0),
    ('res4 3x3 256', 2, 256, 51, 84, 256, 3, 1, 1),
    ('res4 1x1 256->1024', 2, 256, 51, 84, 1024, 1, 1, 0),
    ('res4 1x1 1024->256', 2, 1024, 51, 84, 256, 1, 1, 0),
    ('rpn 3x3 1024', 2, 1024, 51, 84, 1024, 3, 1, 1),
    ('res5a 1x1s2 1024->512', 1024, 1024, 14, 14, 512, 1, 2, 0),
    ('res5a 1x1s2 1024->2048', 1024, 1024, 14, 14, 2048, 1, 2, 0),
    ('res5 3x3 512', 1024, 512, 7, 7, 512, 3, 1, 1),
    ('res5 1x1 512->2048', 1024, 512, 7, 7, 2048, 1, 1, 0),
    ('res5 1x1 2048->512', 1024, 2048, 7, 7, 512, 1, 1, 0),
    ('mask 1x1 256->80', 1024, 256, 14, 14, 80, 1, 1, 0),
    ('fc 2048->408', 1024, 2048, 1, 1, 408, 1, 1, 0),
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


def main():
    only = sys.argv[1] if len(sys.argv) > 1 else None
    lib = _lib.load()
    for kv in os.environ.get('BENCH_TUNE', '').split(','):
        if '=' in kv:
            k, v = kv.split('=')
            _lib.check(lib.mrcnn_set_tuning(k.encode(), int(v)), 'set_tuning')
    tot = {'fwd': 0., 'dgrad': 0., 'wgrad': 0.}
    print('%-28s %11s %11s %11s %11s  (TFLOP/s | ms)' % ('shape', 'fwd', 'dgrad', 'wgrad', 'dgrad_wt'))
    for name, N, C, H, W, K, k, s, p in SHAPES:
        if only and only not in name:
            continue
        x = torch.randn((N, H, W, C), device=dev).permute(0, 3, 1, 2)
        w = (torch.randn((K, k, k, C), device=dev) * 0.05).permute(0, 3, 1, 2)
        d = make_desc(x.shape, w.shape, s, p)
        y = empty_nhwc((d.N, d.K, d.P, d.Q), dev)
        gy = torch.randn((d.N, d.P, d.Q, d.K), device=dev).permute(0, 3, 1, 2)
        gx = empty_nhwc((N, C, H, W), dev)
        gw = torch.empty_like(w)
        ws = _lib.workspace(lib.mrcnn_conv2d_wgrad_workspace_bytes(ctx_desc(d)), dev, 'wgrad')
        sp = _lib.stream_ptr()
        flop = 2.0 * d.N * d.P * d.Q * K * C * k * k
        sw = _lib.ptr(split_ws(dev)) if not os.environ.get('BENCH_NOSPLIT') else None
        f = lambda: _lib.call('mrcnn_conv2d_fwd', ctx_desc(d), _lib.ptr(x), _lib.ptr(w), None, None,
                              None, None, _lib.ptr(y), 0, sw, sp)
        if os.environ.get('BENCH_EPI'):     # affine + residual + ReLU epilogue (a bottleneck's conv3)
            esc = torch.rand((K,), device=dev) + 0.5
            esh = torch.randn((K,), device=dev)
            eres = torch.randn((d.N, d.P, d.Q, d.K), device=dev).permute(0, 3, 1, 2)
            f = lambda: _lib.call('mrcnn_conv2d_fwd', ctx_desc(d), _lib.ptr(x), _lib.ptr(w), None,
                                  _lib.ptr(esc), _lib.ptr(esh), _lib.ptr(eres), _lib.ptr(y),
                                  2 | 4 | 8, sw, sp)
        g = lambda: _lib.call('mrcnn_conv2d_dgrad', ctx_desc(d), _lib.ptr(gy), _lib.ptr(w),
                              _lib.ptr(gx), 0, sp)
        h = lambda: _lib.call('mrcnn_conv2d_wgrad', ctx_desc(d), _lib.ptr(x), _lib.ptr(gy),
                              _lib.ptr(gw), _lib.ptr(ws), sp)
        if os.environ.get('BENCH_MASK'):
            ymask = torch.randn((d.N, d.P, d.Q, d.K), device=dev).permute(0, 3, 1, 2)
            sc = torch.rand((d.K,), device=dev) + 0.5
            g = lambda: _lib.call('mrcnn_conv2d_dgrad_ex', ctx_desc(d), _lib.ptr(gy), _lib.ptr(w),
                                  _lib.ptr(gx), 0, _lib.ptr(ymask), _lib.ptr(sc), None, None, None, None, sw, sp)
            h = lambda: _lib.call('mrcnn_conv2d_wgrad_ex', ctx_desc(d), _lib.ptr(x), _lib.ptr(gy),
                                  _lib.ptr(gw), _lib.ptr(ws), _lib.ptr(ymask), _lib.ptr(sc), None, sp)
        tf, tg, th = timeit(f), timeit(g), timeit(h)
        tot['fwd'] += tf; tot['dgrad'] += tg; tot['wgrad'] += th
        tt = float('nan')
        if s == 1:
            # forward-form dgrad on the flipped/transposed filter, masked + scaled (what a
            # bottleneck backward launches for its stride-1 convolutions)
            ymask = torch.randn((d.N, d.P, d.Q, d.K), device=dev).permute(0, 3, 1, 2)
            sc = torch.rand((d.K,), device=dev) + 0.5
            wT = torch.empty((C * k * k * K,), device=dev)
            _lib.call('mrcnn_filter_flip_transpose', _lib.ptr(w), _lib.ptr(wT), K, k, k, C, None, sp)
            if os.environ.get('BENCH_MASK'):   # consumer-side mask staging
                t = lambda: _lib.call('mrcnn_conv2d_dgrad_wt', ctx_desc(d), _lib.ptr(gy), _lib.ptr(wT),
                                      _lib.ptr(gx), 0, _lib.ptr(ymask), _lib.ptr(sc), None, None,
                                      None, None, sw, sp)
            else:                              # producer-side: mask + scale in the epilogue
                xm = torch.randn((N, H, W, C), device=dev).permute(0, 3, 1, 2)
                sc = torch.rand((C,), device=dev) + 0.5
                t = lambda: _lib.call('mrcnn_conv2d_dgrad_wt', ctx_desc(d), _lib.ptr(gy), _lib.ptr(wT),
                                      _lib.ptr(gx), 0, None, None, None, None, _lib.ptr(xm),
                                      _lib.ptr(sc), sw, sp)
            tt = timeit(t)
        print('%-28s %5.1f|%5.2f %5.1f|%5.2f %5.1f|%5.2f %5.1f|%5.2f' % (
            name, flop / tf / 1e9, tf, flop / tg / 1e9, tg, flop / th / 1e9, th,
            flop / tt / 1e9, tt))
    print('sum ms', tot)


if __name__ == '__main__':
    main()
