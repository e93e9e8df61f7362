"""Developer experiment (historical): the hybrid MFMA + VALU launch — the last `valu_rows_pct`
percent of a big forward-form launch's rows computed by a vector-ALU variant of the GEMM kernel
on a second stream, beside the MFMA launch.  Result (profiles/r02_exp_hybrid_valu_mfma*.{log,txt}):
values identical to fp32 rounding, the VALU-only variant reaches 86 TFLOP/s, but co-resident the
two are zero-sum (MFMA 112 + VALU 18 = the 130 TFLOP/s of the MFMA kernel alone).  The variant
was removed from csrc/conv_gemm.hip again; it lives in the commit "Experiment: VALU variant of
the 128x128 GEMM beside the MFMA launch".  This script needs that commit's library."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from chainer_mask_rcnn_amd import _lib
from chainer_mask_rcnn_amd.functions.conv import make_desc, ctx_desc, split_ws
from chainer_mask_rcnn_amd.functions._layout import empty_nhwc

dev = torch.device('cuda:0')
lib = _lib.load()
SHAPES = [('res5 3x3 512', 1024, 512, 7, 7, 512, 3, 1), ('res5 1x1 512->2048', 1024, 512, 7, 7, 2048, 1, 0),
          ('res5 1x1 2048->512', 1024, 2048, 7, 7, 512, 1, 0), ('rpn 3x3 1024', 2, 1024, 51, 84, 1024, 3, 1)]
pcts = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else '0,10,15,20,25,30').split(',')]
for name, N, C, H, W, K, k, p in SHAPES:
    x = torch.randn((N, H, W, C), device=dev).permute(0, 3, 1, 2)
    w = (torch.randn((K, k, k, C), device=dev) * 0.05).permute(0, 3, 1, 2)
    sc = torch.rand((K,), device=dev) + 0.5
    sh = torch.randn((K,), device=dev)
    d = make_desc(x.shape, w.shape, 1, p)
    res = torch.randn((d.N, d.P, d.Q, d.K), device=dev).permute(0, 3, 1, 2)
    flop = 2.0 * d.N * d.P * d.Q * K * C * k * k
    outs = {}
    for pct in pcts:
        lib.mrcnn_set_tuning(b'valu_rows_pct', pct)
        y = empty_nhwc((d.N, d.K, d.P, d.Q), dev)
        f = lambda: _lib.call('mrcnn_conv2d_fwd', ctx_desc(d), _lib.ptr(x), _lib.ptr(w), None,
                              _lib.ptr(sc), _lib.ptr(sh), _lib.ptr(res), _lib.ptr(y), 2 | 4 | 8,
                              _lib.ptr(split_ws(dev)), _lib.stream_ptr())
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            f()
        b.record()
        torch.cuda.synchronize()
        t = a.elapsed_time(b) / 10
        outs[pct] = y.clone()
        err = (outs[pct] - outs[pcts[0]]).abs().max().item() / outs[pcts[0]].abs().max().item()
        print('%-22s valu_rows_pct %2d: %.3f ms  %6.1f TF/s   max rel diff vs first %.2e' % (
            name, pct, t, flop / t / 1e9, err))
lib.mrcnn_set_tuning(b'valu_rows_pct', 0)
