"""Developer tool: the split-operand (3 x bf16) forward-form kernel next to the fp32 MFMA kernel —
error of both against float64 on the same inputs, and TFLOP/s, per shape."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from chainer_mask_rcnn_amd import _lib
from chainer_mask_rcnn_amd.functions.conv import make_desc, ctx_desc, split_ws
from chainer_mask_rcnn_amd.functions._layout import empty_nhwc

dev = torch.device('cuda:0')
SHAPES = [('res5 1x1 512->2048', 1024, 512, 7, 7, 2048, 1, 1, 0),
          ('res5 1x1 2048->512', 1024, 2048, 7, 7, 512, 1, 1, 0),
          ('res5 1x1 1024->2048 (14x14 s1)', 256, 1024, 14, 14, 2048, 1, 1, 0),
          ('res5 3x3 512', 1024, 512, 7, 7, 512, 3, 1, 1),
          ('res4 3x3 256', 2, 256, 51, 84, 256, 3, 1, 1),
          ('rpn 3x3 1024', 2, 1024, 51, 84, 1024, 3, 1, 1)]


if os.environ.get('CHECK_SHAPES'):      # "name:N:C:H:W:K:k:s:p;..."
    SHAPES = [tuple([f.split(':')[0]] + [int(v) for v in f.split(':')[1:]])
              for f in os.environ['CHECK_SHAPES'].split(';')]


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
    lib = _lib.load()
    torch.manual_seed(0)
    for name, N, C, H, W, K, k, s, p in SHAPES:
        x = torch.randn((N, H, W, C), device=dev).permute(0, 3, 1, 2)
        w = (torch.randn((K, k, k, C), device=dev) * 0.05).permute(0, 3, 1, 2)
        d = make_desc(x.shape, w.shape, s, p)
        sw = _lib.ptr(split_ws(dev))
        sp = _lib.stream_ptr()
        flop = 2.0 * d.N * d.P * d.Q * K * C * k * k
        out = {}
        for mode in (0, 1):
            _lib.check(lib.mrcnn_set_tuning(b'split_bf16', mode), 'tune')
            y = empty_nhwc((d.N, d.K, d.P, d.Q), dev)
            f = lambda: _lib.call('mrcnn_conv2d_fwd', ctx_desc(d), _lib.ptr(x), _lib.ptr(w), None, None,
                                  None, None, _lib.ptr(y), 0, sw, sp)
            f()
            torch.cuda.synchronize()
            out[mode] = (y.clone(), timeit(f))
        _lib.check(lib.mrcnn_set_tuning(b'split_bf16', 0), 'tune')
        # float64 reference on a sample of images (all channels)
        ns = min(N, 8)
        ref = torch.nn.functional.conv2d(x[:ns].double(), w.double(), None, s, p)
        scale = ref.abs().max().item()
        line = '%-32s' % name
        for mode in (0, 1):
            err = (out[mode][0][:ns].double() - ref).abs()
            line += '  %s: %6.1f TF  max|err|/max|y| %.2e  rms %.2e' % (
                'split' if mode else 'fp32 ', flop / out[mode][1] / 1e9, err.max().item() / scale,
                err.pow(2).mean().sqrt().item() / scale)
        line += '  max|split - fp32|/max|y| %.2e' % ((out[0][0] - out[1][0]).abs().max().item() / scale)
        print(line, flush=True)
        # weight gradient
        gy = torch.randn((d.N, d.P, d.Q, d.K), device=dev).permute(0, 3, 1, 2)
        ws = _lib.workspace(lib.mrcnn_conv2d_wgrad_workspace_bytes(ctx_desc(d)), dev, 'wgrad')
        out = {}
        for mode in (0, 1):
            _lib.check(lib.mrcnn_set_tuning(b'split_bf16', mode), 'tune')
            gw = torch.empty_like(w)
            h = lambda: _lib.call('mrcnn_conv2d_wgrad', ctx_desc(d), _lib.ptr(x), _lib.ptr(gy),
                                  _lib.ptr(gw), _lib.ptr(ws), sp)
            h()
            torch.cuda.synchronize()
            out[mode] = (gw.clone(), timeit(h))
        _lib.check(lib.mrcnn_set_tuning(b'split_bf16', 0), 'tune')
        kk = min(K, 64)
        ref = torch.nn.grad.conv2d_weight(x.double(), (kk, C, k, k), gy[:, :kk].double(), s, p) if N <= 8 else None
        if ref is None:      # many small maps: float64 reference on the first output channels via einsum
            xp = torch.nn.functional.unfold(x.double(), k, padding=p, stride=s)      # (N, C*k*k, P*Q)
            ref = torch.einsum('nkp,ncp->kc', gy[:, :kk].double().reshape(N, kk, -1), xp).reshape(kk, C, k, k)
        scale = ref.abs().max().item()
        line = '%-32s' % ('  wgrad')
        for mode in (0, 1):
            err = (out[mode][0][:kk].double() - ref).abs()
            line += '  %s: %6.1f TF  max|err|/max|y| %.2e  rms %.2e' % (
                'split' if mode else 'fp32 ', flop / out[mode][1] / 1e9, err.max().item() / scale,
                err.pow(2).mean().sqrt().item() / scale)
        print(line, flush=True)


if __name__ == '__main__':
    main()
