"""Developer experiment: does vector-ALU fp32 work run beside the fp32-MFMA GEMM for free?"""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from chainer_mask_rcnn_amd import _lib
from chainer_mask_rcnn_amd.functions.conv import make_desc, ctx_desc, split_ws
from chainer_mask_rcnn_amd.functions._layout import empty_nhwc

dev = torch.device('cuda:0')
lib = _lib.load()
exp = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libexp.so'))
exp.exp_launch_valu.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]

N, C, H, W, K, k = 1024, 512, 7, 7, 512, 3
x = torch.randn((N, H, W, C), device=dev).permute(0, 3, 1, 2)
w = (torch.randn((K, k, k, C), device=dev) * 0.05).permute(0, 3, 1, 2)
d = make_desc(x.shape, w.shape, 1, 1)
y = empty_nhwc((d.N, d.K, d.P, d.Q), dev)
flop = 2.0 * d.N * d.P * d.Q * K * C * k * k
out = torch.zeros(1 << 22, device=dev)
s_main = torch.cuda.current_stream()
s_side = torch.cuda.Stream()


def gemm(reps=4):
    for _ in range(reps):
        _lib.call('mrcnn_conv2d_fwd', ctx_desc(d), _lib.ptr(x), _lib.ptr(w), None, None, None, None,
                  _lib.ptr(y), 0, _lib.ptr(split_ws(dev)), ctypes.c_void_p(s_main.cuda_stream))


def valu(blocks, iters, stream):
    exp.exp_launch_valu(ctypes.c_void_p(out.data_ptr()), blocks, iters, ctypes.c_void_p(stream.cuda_stream))


def timed(fn_main, fn_side=None):
    torch.cuda.synchronize()
    a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if fn_side is not None:
        b0.record(s_side)
    a0.record(s_main)
    fn_main()
    if fn_side is not None:
        fn_side()
        b1.record(s_side)
    a1.record(s_main)
    torch.cuda.synchronize()
    return a0.elapsed_time(a1), (b0.elapsed_time(b1) if fn_side is not None else None)


def valu_flops(blocks, iters):
    return blocks * 256 * iters * 16 * 2.0


for extra in (0, 24000):
    lib.mrcnn_set_tuning(b'gemm_extra_lds', extra)
    gemm(2)
    t, _ = timed(lambda: gemm(4))
    print('GEMM alone, extra LDS %5d: %.3f ms per launch, %.1f TF/s' % (extra, t / 4, flop * 4 / t / 1e9))
for wg_per_cu in (1, 2, 4, 8):
    blocks, iters = 256 * wg_per_cu, 200000 // wg_per_cu
    valu(blocks, 1000, s_main)
    t, _ = timed(lambda: valu(blocks, iters, s_main))
    print('VALU alone, %d WG/CU: %.3f ms, %.1f TF/s' % (wg_per_cu, t, valu_flops(blocks, iters) / t / 1e9))
for extra in (0, 24000):
    lib.mrcnn_set_tuning(b'gemm_extra_lds', extra)
    for wg_per_cu in (1, 2, 4):
        blocks, iters = 256 * wg_per_cu, 260000 // wg_per_cu
        tg, tv = timed(lambda: gemm(4), lambda: valu(blocks, iters, s_side))
        print('together, extra LDS %5d, VALU %d WG/CU: GEMM %.3f ms per launch (%.1f TF/s), VALU %.3f ms '
              '(%.1f TF/s) -> combined %.1f TF/s over the GEMM window' % (
                  extra, wg_per_cu, tg / 4, flop * 4 / tg / 1e9, tv, valu_flops(blocks, iters) / tv / 1e9,
                  (flop * 4 + valu_flops(blocks, iters) * min(1.0, tg / tv)) / tg / 1e9))
lib.mrcnn_set_tuning(b'gemm_extra_lds', 0)
