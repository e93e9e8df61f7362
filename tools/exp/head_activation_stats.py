"""Experiment: activation statistics at the inputs of the RoI head's 3x3 layers in the random-init
test models, and the Winograd-vs-direct forward difference there."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from test_gpu_model import _build
from chainer_mask_rcnn_amd.functions import conv as C
C.WINOGRAD_MIN_WORK = 0
dev = torch.device('cuda:0')
orig = C._fwd_raw


def spy(x, Wc, d, scale, shift, residual, relu):
    y = orig(x, Wc, d, scale, shift, residual, relu)
    if d.R == 3 and d.N >= 32 and d.H == 7:
        yw, _ = C.wino_fwd(x, Wc, d, scale, shift, relu)
        xa = x.abs()
        pre = orig(x, Wc, d, None, None, None, False)
        diff = (yw - y).abs()
        print('  conv2 N=%d: input max %.3g rms %.3g (max/rms %.0f)  |pre| max %.3g rms %.3g   wino-direct: max %.2e rms %.2e of output max %.3g (rel-to-rms %.2e)' % (
            d.N, xa.max().item(), x.pow(2).mean().sqrt().item(), xa.max().item() / x.pow(2).mean().sqrt().item(),
            pre.abs().max().item(), pre.pow(2).mean().sqrt().item(), diff.max().item(), diff.pow(2).mean().sqrt().item(),
            y.abs().max().item(), diff.pow(2).mean().sqrt().item() / y.pow(2).mean().sqrt().item()))
    return y


C._fwd_raw = spy
for n in (50, 101):
    print('R-%d' % n)
    model, chain, imgs, bboxes, labels, masks = _build(dev, n)
    np.random.seed(123)
    chain(torch.tensor(imgs, device=dev), bboxes, labels, masks, [1., 1.])
