"""Developer check (historical — needs the library of the commit "Experiment: LDS-DMA ...";
the variant was removed from csrc/conv_gemm.hip again, results in profiles/r02_exp_lds_dma.log):
the LDS-DMA variant of the forward-form 128x128 kernel (mrcnn_set_tuning
"lds_dma") against the register-staged kernel — bit-identical outputs expected (same MFMA
order; only the staging differs) — on shapes that exercise padding taps, channel tails,
position-major rows and the fused epilogue; plus timing."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from chainer_mask_rcnn_amd import _lib
from chainer_mask_rcnn_amd.functions.conv import make_desc, ctx_desc, split_ws
from chainer_mask_rcnn_amd.functions._layout import empty_nhwc

dev = torch.device('cuda:0')
lib = _lib.load()
SHAPES = [  # name, N, C, H, W, K, k, pad
    ('res5 3x3 512', 1024, 512, 7, 7, 512, 3, 1), ('res5 1x1 512->2048', 1024, 512, 7, 7, 2048, 1, 0),
    ('res5 1x1 2048->512', 1024, 2048, 7, 7, 512, 1, 0), ('rpn 3x3 1024', 2, 1024, 51, 84, 1024, 3, 1),
    ('C tail 36', 2, 36, 120, 130, 256, 3, 1), ('K tail 132', 3, 128, 90, 100, 132, 3, 1),
    ('res2 1x1 64->256', 2, 64, 201, 334, 256, 1, 0), ('perm 300 imgs', 300, 64, 7, 7, 256, 3, 1)]
torch.manual_seed(0)
for name, N, C, H, W, K, k, p in SHAPES:
    x = torch.randn((N, H, W, C), device=dev).permute(0, 3, 1, 2)
    w = (torch.randn((K, k, k, C), device=dev) * 0.05).permute(0, 3, 1, 2)
    sc = torch.rand((K,), device=dev) + 0.5
    sh = torch.randn((K,), device=dev)
    d = make_desc(x.shape, w.shape, 1, p)
    res = torch.randn((d.N, d.P, d.Q, d.K), device=dev).permute(0, 3, 1, 2)
    flop = 2.0 * d.N * d.P * d.Q * K * C * k * k
    outs, times = {}, {}
    for dma in (0, 1, 0, 1):
        lib.mrcnn_set_tuning(b'lds_dma', dma)
        y = empty_nhwc((d.N, d.K, d.P, d.Q), dev)
        y.fill_(float('nan'))
        f = lambda: _lib.call('mrcnn_conv2d_fwd', ctx_desc(d), _lib.ptr(x), _lib.ptr(w), None,
                              _lib.ptr(sc), _lib.ptr(sh), _lib.ptr(res), _lib.ptr(y), 2 | 4 | 8,
                              _lib.ptr(split_ws(dev)), _lib.stream_ptr())
        for _ in range(2):
            f()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(8):
            f()
        b.record()
        torch.cuda.synchronize()
        times.setdefault(dma, []).append(a.elapsed_time(b) / 8)
        outs[dma] = y.clone()
    same = torch.equal(outs[0], outs[1])
    err = (outs[0] - outs[1]).abs().max().item()
    print('%-20s reg %.3f ms %6.1f TF/s | dma %.3f ms %6.1f TF/s | identical %s (max abs diff %.2e, nan %d)' % (
        name, min(times[0]), flop / min(times[0]) / 1e9, min(times[1]), flop / min(times[1]) / 1e9, same, err,
        int(torch.isnan(outs[1]).sum())))
lib.mrcnn_set_tuning(b'lds_dma', 0)
