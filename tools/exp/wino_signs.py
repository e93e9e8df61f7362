"""Calibration of MRCNN_EPI_EXACT_SIGNS: for the head / RPN shapes, how many ReLU decisions of the
Winograd forward differ from the float64 truth, with and without the fix-up, how many outputs the
fix-up recomputes, and the largest |pre-activation| (fp64) among the remaining disagreements."""
import os, sys, ctypes
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from chainer_mask_rcnn_amd import _lib
from chainer_mask_rcnn_amd.functions import conv as C
from chainer_mask_rcnn_amd.functions._layout import nhwc

dev = torch.device('cuda:0')
lib = _lib.load()


def truth(x, w, sc, sh):
    y = torch.nn.functional.conv2d(x.double(), w.double(), None, 1, 1)
    return y * sc.double()[None, :, None, None] + sh.double()[None, :, None, None]


def run(N, Cc, K, H, W, seed, ppbs):
    g = torch.Generator(device='cpu').manual_seed(seed)
    x = torch.randn(N, Cc, H, W, generator=g).relu_().mul_(20.)
    w = torch.randn(K, Cc, 3, 3, generator=g) / (3. * Cc ** 0.5)
    sc = torch.rand(K, generator=g) * 0.5 + 0.4
    sh = torch.randn(K, generator=g) * 2.0
    pre = truth(x, w, sc, sh).to(dev)            # computed on the CPU in float64
    scale = pre.abs().max().item()
    xt, wt, sct, sht = nhwc(x.to(dev)), nhwc(w.to(dev)), sc.to(dev), sh.to(dev)
    d = C.make_desc(xt.shape, wt.shape, 1, 1)
    yd = C._fwd_raw(xt, wt, d, sct, sht, None, True)
    def disagreements(y):
        bad = (y > 0) != (pre > 0)
        n = int(bad.sum())
        worst = (pre.abs()[bad].max().item() / scale) if n else 0.
        return n, worst
    total = pre.numel()
    print('shape N=%d C=%d K=%d %dx%d: %d outputs, |pre| max %.3g' % (N, Cc, K, H, W, total, scale))
    print('   direct kernel      : %4d sign disagreements with fp64 (largest |pre|/scale %.1e)' % disagreements(yd))
    yw, _ = C.wino_fwd(xt, wt, d, sct, sht, True)
    print('   winograd           : %4d sign disagreements (largest %.1e); max |y - direct| / scale %.1e' % (
        disagreements(yw) + (((yw - yd).abs().max().item()) / scale,)))
    for ppb in ppbs:
        _lib.check(lib.mrcnn_set_tuning(b'wino_ambiguity_ppb', ppb), 'tune')
        yf, _ = C.wino_fwd(xt, wt, d, sct, sht, True, exact_signs=True)
        cnt = ctypes.c_int(0)
        _lib.call('mrcnn_conv3x3_wino_fixup_count', C.ctx_desc(d), _lib.ptr(C._wino_ws(d, dev)),
                  _lib.stream_ptr(), ctypes.byref(cnt))
        n, worst = disagreements(yf)
        print('   + fix-up %5d ppb  : %4d sign disagreements (largest %.1e); recomputed %d outputs (%.1e of all)' % (
            ppb, n, worst, cnt.value, cnt.value / total))


ppbs = (250, 1000, 4000, 16000)
run(512, 512, 512, 7, 7, 0, ppbs)
run(2, 1024, 1024, 50, 84, 1, ppbs)
run(1024, 512, 512, 7, 7, 2, ppbs)
