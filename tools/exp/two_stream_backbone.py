"""Developer experiment: the batch-2 backbone as ONE chain of batch-2 launches vs TWO concurrent
chains of batch-1 launches on two streams (do the under-filled small-M launches fill each
other's tails?).  One res4-style bottleneck chain x 6 blocks, forward only."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from chainer_mask_rcnn_amd import _lib
from chainer_mask_rcnn_amd.functions.conv import make_desc, ctx_desc
from chainer_mask_rcnn_amd.functions._layout import empty_nhwc
import ctypes

dev = torch.device('cuda:0')
lib = _lib.load()
H, W = 51, 84
ws_bytes = lib.mrcnn_conv2d_split_workspace_bytes()


def make(n):
    x = torch.randn((n, H, W, 1024), device=dev).permute(0, 3, 1, 2)
    w1 = (torch.randn((256, 1, 1, 1024), device=dev) * 0.03).permute(0, 3, 1, 2)
    w2 = (torch.randn((256, 3, 3, 256), device=dev) * 0.03).permute(0, 3, 1, 2)
    w3 = (torch.randn((1024, 1, 1, 256), device=dev) * 0.03).permute(0, 3, 1, 2)
    d1 = make_desc(x.shape, w1.shape, 1, 0)
    h1 = empty_nhwc((n, 256, H, W), dev)
    d2 = make_desc(h1.shape, w2.shape, 1, 1)
    h2 = empty_nhwc((n, 256, H, W), dev)
    d3 = make_desc(h2.shape, w3.shape, 1, 0)
    y = empty_nhwc((n, 1024, H, W), dev)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    return dict(x=x, w=(w1, w2, w3), d=(d1, d2, d3), h=(h1, h2, y), ws=ws)


def chain(c, stream, blocks=6):
    sp = ctypes.c_void_p(stream.cuda_stream)
    for _ in range(blocks):
        src = c['x']
        for w, d, dst in zip(c['w'], c['d'], c['h']):
            _lib.call('mrcnn_conv2d_fwd', ctx_desc(d), _lib.ptr(src), _lib.ptr(w), None, None, None, None,
                      _lib.ptr(dst), 8, _lib.ptr(c['ws']), sp)
            src = dst


main = torch.cuda.current_stream()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
full, a, b = make(2), make(1), make(1)
flop = 6 * 2.0 * 2 * H * W * (1024 * 256 + 256 * 256 * 9 + 256 * 1024)


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(main)
    for _ in range(reps):
        fn()
    e1.record(main)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def one():
    chain(full, main)


def two():
    s1.wait_stream(main); s2.wait_stream(main)
    chain(a, s1); chain(b, s2)
    main.wait_stream(s1); main.wait_stream(s2)


for name, fn in (('one chain, batch 2', one), ('two chains, batch 1 each, two streams', two),
                 ('one chain, batch 2', one), ('two chains, batch 1 each, two streams', two)):
    t = timed(fn)
    print('%-42s %.3f ms  %.1f TF/s' % (name, t, flop / t / 1e9))
