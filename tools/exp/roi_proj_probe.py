"""Developer probe: the ROIAlign kernels of the projected head in isolation — forward with the affine
epilogue and pixel-owner backward on the 512- and 2048-channel maps of the C2 shape — on two RoI
populations (the tiny proposals of a random-init RPN, object-sized boxes), for a list of lane counts
per workgroup (mrcnn_set_tuning roi_fwd_lanes / roi_bwd_lanes).  Prints us and algorithmic TB/s."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from chainer_mask_rcnn_amd import _lib
from chainer_mask_rcnn_amd.functions import conv as C
from chainer_mask_rcnn_amd.functions.roi_align_2d import spatial_order
from chainer_mask_rcnn_amd.functions._layout import nhwc

dev = torch.device('cuda:0')
N, H, W, R = 2, 51, 84, 1024
rb = np.random.RandomState(5)


def boxes(kind):
    if kind == 'tiny':
        hh, ww = rb.uniform(48, 160, R), rb.uniform(48, 160, R)
    else:
        hh, ww = rb.uniform(32, 600, R), rb.uniform(32, 600, R)
    y0, x0 = rb.uniform(0, 800 - 32, R), rb.uniform(0, 1333 - 32, R)
    b = np.stack([y0, x0, np.minimum(y0 + hh, 800), np.minimum(x0 + ww, 1333)], 1).astype(np.float32)
    idx = np.repeat(np.arange(N), R // N).astype(np.int32)
    return b, idx


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        torch.cuda._sleep(300000)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return float(np.median(ts))


lanes = [int(v) for v in os.environ.get('LANES', '256,128,64').split(',')]
for kind in ('tiny', 'object'):
    b, idx = boxes(kind)
    order = torch.tensor(spatial_order(b, idx, 1 / 16.), device=dev)
    r5 = torch.tensor(np.concatenate([idx[:, None].astype(np.float32), b[:, [1, 0, 3, 2]]], 1), device=dev)
    spec = C.RoiSpec(r5, 14, 14, 1 / 16., bin_stride=2, order=order)
    for Cn in (512, 2048):
        z = nhwc(torch.randn((N, Cn, H, W), device=dev))
        sc, sh = torch.rand(Cn, device=dev) + 0.5, torch.randn(Cn, device=dev)
        gy = nhwc(torch.randn((R, Cn, 7, 7), device=dev))
        byts = 4.0 * (R * 49 * Cn + N * H * W * Cn)
        for ln in lanes:
            _lib.set_tuning('roi_fwd_lanes', ln); _lib.set_tuning('roi_bwd_lanes', ln)
            tf = timed(lambda: C._roi_pool_affine(z, spec, sc, sh, True))
            tb = timed(lambda: C._roi_pool_bwd(gy, spec, (N, Cn, H, W)))
            print('%-6s C=%4d lanes %3d  fwd %6.1f us %.2f TB/s (%.3f)   bwd %6.1f us %.2f TB/s (%.3f)' % (
                kind, Cn, ln, tf, byts / tf / 1e6, byts / tf / 1e6 / 8, tb, byts / tb / 1e6, byts / tb / 1e6 / 8))
