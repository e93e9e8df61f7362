"""Loss / optimizer kernels vs the NumPy oracle."""
import numpy as np
import pytest
import torch

from oracle import np_ref
from chainer_mask_rcnn_amd import functions as F
from chainer_mask_rcnn_amd import _lib

pytestmark = pytest.mark.gpu


def _t(a, dev, grad=False):
    t = torch.tensor(a, device=dev)
    if grad:
        t.requires_grad_(True)
    return t


@pytest.mark.parametrize('n', [1, 300, 128520])
def test_sigmoid_cross_entropy(dev, n):
    rng = np.random.RandomState(n)
    x = (rng.standard_normal(n) * 3).astype(np.float32)
    t = rng.randint(-1, 2, n).astype(np.int32)
    xt = _t(x, dev, True)
    loss = F.sigmoid_cross_entropy(xt, _t(t, dev))
    loss.backward()
    l_ref, g_ref = np_ref.sigmoid_cross_entropy(x, t)
    np.testing.assert_allclose(loss.item(), l_ref, rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(xt.grad.cpu().numpy(), g_ref, rtol=1e-4, atol=1e-8)


def test_sigmoid_cross_entropy_all_ignored(dev):
    xt = _t(np.ones(10, np.float32), dev, True)
    loss = F.sigmoid_cross_entropy(xt, _t(np.full(10, -1, np.int32), dev))
    loss.backward()
    assert loss.item() == 0. and float(xt.grad.abs().sum()) == 0.


def test_mask_sigmoid_cross_entropy(dev):
    rng = np.random.RandomState(1)
    R, Kc, M = 64, 80, 14
    x = rng.standard_normal((R, Kc, M, M)).astype(np.float32)
    label = rng.randint(0, Kc + 1, R).astype(np.int32)      # 0 = background
    t = rng.randint(0, 2, (R, M, M)).astype(np.int32)
    t[label == 0] = -1
    xt = _t(x, dev, True)
    loss = F.mask_sigmoid_cross_entropy(xt, _t(label, dev), _t(t, dev))
    loss.backward()
    sel = x[np.arange(R), label - 1]
    l_ref, g_sel = np_ref.sigmoid_cross_entropy(sel, t)
    g_ref = np.zeros_like(x)
    g_ref[np.arange(R), label - 1] = g_sel
    np.testing.assert_allclose(loss.item(), l_ref, rtol=1e-4)
    np.testing.assert_allclose(xt.grad.cpu().numpy(), g_ref, rtol=1e-4, atol=1e-8)


def test_softmax_cross_entropy_and_softmax(dev):
    rng = np.random.RandomState(2)
    x = (rng.standard_normal((1024, 81)) * 2).astype(np.float32)
    t = rng.randint(-1, 81, 1024).astype(np.int32)
    xt = _t(x, dev, True)
    loss = F.softmax_cross_entropy(xt, _t(t, dev))
    loss.backward()
    l_ref, g_ref = np_ref.softmax_cross_entropy(x, t)
    np.testing.assert_allclose(loss.item(), l_ref, rtol=1e-4)
    np.testing.assert_allclose(xt.grad.cpu().numpy(), g_ref, rtol=1e-4, atol=1e-8)
    # strided view (fused head output)
    big = np.zeros((1024, 408), np.float32)
    big[:, 324:405] = x
    p = F.softmax(_t(big, dev)[:, 324:405])
    e = np.exp(x - x.max(1, keepdims=True))
    np.testing.assert_allclose(p.cpu().numpy(), e / e.sum(1, keepdims=True), rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize('sigma', [1., 3.])
def test_fast_rcnn_loc_loss(dev, sigma):
    rng = np.random.RandomState(3)
    n = 5000
    pred = rng.standard_normal((n, 4)).astype(np.float32)
    gt = rng.standard_normal((n, 4)).astype(np.float32)
    label = rng.randint(-1, 3, n).astype(np.int32)
    pt = _t(pred, dev, True)
    loss = F.fast_rcnn_loc_loss(pt, _t(gt, dev), _t(label, dev), sigma)
    loss.backward()
    l_ref, g_ref = np_ref.fast_rcnn_loc_loss(pred, gt, label, sigma)
    np.testing.assert_allclose(loss.item(), l_ref, rtol=1e-4)
    np.testing.assert_allclose(pt.grad.cpu().numpy(), g_ref, rtol=1e-4, atol=1e-8)


def test_fast_rcnn_loc_loss_class_select(dev):
    rng = np.random.RandomState(4)
    n, ncls = 512, 81
    pred = rng.standard_normal((n, ncls * 4)).astype(np.float32)
    gt = rng.standard_normal((n, 4)).astype(np.float32)
    label = rng.randint(0, ncls, n).astype(np.int32)
    pt = _t(pred, dev, True)
    loss = F.fast_rcnn_loc_loss(pt, _t(gt, dev), _t(label, dev), 1., cls=_t(label, dev))
    loss.backward()
    sel = pred.reshape(n, ncls, 4)[np.arange(n), label]
    l_ref, g_sel = np_ref.fast_rcnn_loc_loss(sel, gt, label, 1.)
    g_ref = np.zeros((n, ncls, 4), np.float32)
    g_ref[np.arange(n), label] = g_sel
    np.testing.assert_allclose(loss.item(), l_ref, rtol=1e-4)
    np.testing.assert_allclose(pt.grad.cpu().numpy(), g_ref.reshape(n, -1), rtol=1e-4, atol=1e-8)


def test_sgd_momentum_wd(dev):
    rng = np.random.RandomState(5)
    n = 1000003
    p = rng.standard_normal(n).astype(np.float32)
    g = rng.standard_normal(n).astype(np.float32)
    v = rng.standard_normal(n).astype(np.float32)
    pt, gt, vt = _t(p, dev), _t(g, dev), _t(v, dev)
    _lib.call('mrcnn_sgd_momentum_wd', _lib.ptr(pt), _lib.ptr(gt), _lib.ptr(vt), n,
              0.02, 0.9, 1e-4, 1.0, _lib.stream_ptr())
    p2, v2 = np_ref.momentum_sgd_wd(p, g, v, 0.02)
    np.testing.assert_allclose(pt.cpu().numpy(), p2, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(vt.cpu().numpy(), v2, rtol=1e-5, atol=1e-6)


def test_loc_loss_matches_reference_function_fixture(dev, golden_dir):
    import os
    d = np.load(os.path.join(golden_dir, 'loc_loss.npz'))
    for sigma, key in ((3., 'loss_sigma3'), (1., 'loss_sigma1')):
        loss = F.fast_rcnn_loc_loss(_t(d['pred'], dev), _t(d['gt'], dev), _t(d['label'], dev), sigma)
        np.testing.assert_allclose(loss.item(), float(d[key]), rtol=1e-5)
