"""Developer prototype driver: tools/exp/dma_gemm_probe.hip (plane images + LDS-DMA staging) against
the library's split-operand forward-form kernel on the RoI head's 1x1 shapes — bit equality and
TFLOP/s at the power cap."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from chainer_mask_rcnn_amd import _lib
from chainer_mask_rcnn_amd.functions.conv import make_desc, ctx_desc, split_ws
from chainer_mask_rcnn_amd.functions._layout import empty_nhwc

dev = torch.device('cuda:0')
here = os.path.dirname(os.path.abspath(__file__))
probe = ctypes.CDLL(os.path.join(here, 'libdma_gemm_probe.so'))
probe.dma_gemm.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3 + [ctypes.c_void_p]


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def planes_of(t, rows, L):
    pl = torch.empty((rows * L * 6,), dtype=torch.uint8, device=dev)
    _lib.call('mrcnn_split_planes', _lib.ptr(t), _lib.ptr(pl), rows, L, _lib.stream_ptr())
    return pl


def main():
    _lib.load()
    for name, N, Cc, K in (('res5 1x1 2048->512', 1024, 2048, 512), ('res5 1x1 512->2048', 1024, 512, 2048),
                           ('res5a 1x1 1024->2048 (binned)', 1024, 1024, 2048), ('wino-like 512->512', 1337, 512, 512)):
        H = W = 7
        x = torch.randn((N, H, W, Cc), device=dev).permute(0, 3, 1, 2)
        w = (torch.randn((K, 1, 1, Cc), device=dev) * 0.05).permute(0, 3, 1, 2)
        d = make_desc(x.shape, w.shape, 1, 0)
        y = empty_nhwc((d.N, d.K, d.P, d.Q), dev)
        M = N * H * W
        x_pl, w_pl = planes_of(x, M, Cc), planes_of(w, K, Cc)
        y2 = torch.zeros((M, K), device=dev)
        sp = _lib.stream_ptr()
        ref = lambda: _lib.call('mrcnn_conv2d_fwd', ctx_desc(d), _lib.ptr(x), _lib.ptr(w), None, None, None,
                                None, _lib.ptr(y), 0, _lib.ptr(split_ws(dev)), sp)
        new = lambda: probe.dma_gemm(_lib.ptr(x_pl), _lib.ptr(w_pl), _lib.ptr(y2), M, K, Cc, sp)
        ref(); rc = new(); torch.cuda.synchronize()
        same = torch.equal(y.permute(0, 2, 3, 1).reshape(M, K), y2)
        err = (y.permute(0, 2, 3, 1).reshape(M, K) - y2).abs().max().item()
        flop = 2.0 * M * K * Cc
        tr, tn = timeit(ref), timeit(new)
        print('%-32s library %6.1f TF (%.3f ms) | planes + LDS-DMA %6.1f TF (%.3f ms) | rc %d bit-identical %s (max diff %.2e)'
              % (name, flop / tr / 1e9, tr, flop / tn / 1e9, tn, rc, same, err))


if __name__ == '__main__':
    main()
