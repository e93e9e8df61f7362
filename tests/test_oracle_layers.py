"""Pin oracle/np_ref.py's restatements of the chainer layers and losses against an
independent fp32 implementation (torch.nn.functional on CPU).  The reference pins
nothing here ("parity unpinned")."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import np_ref

RT, AT = 1e-4, 1e-4


@pytest.mark.parametrize('k,s,p', [(1, 1, 0), (3, 1, 1), (1, 2, 0), (7, 2, 3), (2, 2, 0)])
def test_conv2d_fwd_bwd(k, s, p):
    rng = np.random.RandomState(0)
    x = rng.standard_normal((2, 5, 13, 17)).astype(np.float32)
    W = rng.standard_normal((6, 5, k, k)).astype(np.float32)
    b = rng.standard_normal(6).astype(np.float32)
    y = np_ref.conv2d_fwd(x, W, b, s, p)
    xt = torch.tensor(x, requires_grad=True)
    Wt = torch.tensor(W, requires_grad=True)
    bt = torch.tensor(b, requires_grad=True)
    yt = F.conv2d(xt, Wt, bt, stride=s, padding=p)
    np.testing.assert_allclose(y, yt.detach().numpy(), rtol=RT, atol=AT)
    gy = rng.standard_normal(y.shape).astype(np.float32)
    yt.backward(torch.tensor(gy))
    gx, gW, gb = np_ref.conv2d_bwd(x, W, gy, s, p)
    np.testing.assert_allclose(gx, xt.grad.numpy(), rtol=RT, atol=AT)
    np.testing.assert_allclose(gW, Wt.grad.numpy(), rtol=RT, atol=AT * 5)
    np.testing.assert_allclose(gb, bt.grad.numpy(), rtol=RT, atol=AT * 5)


def test_deconv2x2s2():
    rng = np.random.RandomState(1)
    x = rng.standard_normal((3, 6, 5, 4)).astype(np.float32)
    W = rng.standard_normal((6, 7, 2, 2)).astype(np.float32)
    b = rng.standard_normal(7).astype(np.float32)
    y = np_ref.deconv2x2s2_fwd(x, W, b)
    xt = torch.tensor(x, requires_grad=True)
    Wt = torch.tensor(W, requires_grad=True)
    yt = F.conv_transpose2d(xt, Wt, torch.tensor(b), stride=2)
    np.testing.assert_allclose(y, yt.detach().numpy(), rtol=RT, atol=AT)
    gy = rng.standard_normal(y.shape).astype(np.float32)
    yt.backward(torch.tensor(gy))
    gx, gW, gb = np_ref.deconv2x2s2_bwd(x, W, gy)
    np.testing.assert_allclose(gx, xt.grad.numpy(), rtol=RT, atol=AT)
    np.testing.assert_allclose(gW, Wt.grad.numpy(), rtol=RT, atol=AT * 5)


@pytest.mark.parametrize('hw', [(400, 667), (12, 8), (13, 9), (101, 167)])
def test_max_pool_cover_all(hw):
    # chainer's cover_all=True == torch's ceil_mode=True for k3/s2/p1
    rng = np.random.RandomState(2)
    h, w = hw
    x = rng.standard_normal((1, 2, h, w)).astype(np.float32)
    y = np_ref.max_pooling_2d(x)
    yt = F.max_pool2d(torch.tensor(x), 3, 2, 1, ceil_mode=True).numpy()
    assert y.shape == yt.shape
    np.testing.assert_array_equal(y, yt)
    assert y.shape[2] == np_ref.conv_outsize(h, 3, 2, 1, cover_all=True)


def test_c4_shapes():
    # 800x1333 -> 400x667 -> 201x334 -> 101x167 -> 51x84
    h, w = 800, 1333
    h, w = np_ref.conv_outsize(h, 7, 2, 3), np_ref.conv_outsize(w, 7, 2, 3)
    assert (h, w) == (400, 667)
    h, w = np_ref.conv_outsize(h, 3, 2, 1, True), np_ref.conv_outsize(w, 3, 2, 1, True)
    assert (h, w) == (201, 334)
    h, w = np_ref.conv_outsize(h, 1, 2, 0), np_ref.conv_outsize(w, 1, 2, 0)
    assert (h, w) == (101, 167)
    h, w = np_ref.conv_outsize(h, 1, 2, 0), np_ref.conv_outsize(w, 1, 2, 0)
    assert (h, w) == (51, 84)


def test_avg_pool_linear():
    rng = np.random.RandomState(3)
    x = rng.standard_normal((4, 6, 7, 7)).astype(np.float32)
    y = np_ref.average_pooling_2d(x, 7, 7)
    np.testing.assert_allclose(y[:, :, 0, 0], x.mean(axis=(2, 3)), rtol=1e-5, atol=1e-6)
    W = rng.standard_normal((5, 6)).astype(np.float32)
    b = rng.standard_normal(5).astype(np.float32)
    out = np_ref.linear_fwd(y, W, b)
    ref = F.linear(torch.tensor(y.reshape(4, 6)), torch.tensor(W), torch.tensor(b)).numpy()
    np.testing.assert_allclose(out, ref, rtol=RT, atol=AT)


def test_sigmoid_cross_entropy():
    rng = np.random.RandomState(4)
    x = (rng.standard_normal(500) * 3).astype(np.float32)
    t = rng.randint(-1, 2, 500).astype(np.int32)
    loss, gx = np_ref.sigmoid_cross_entropy(x, t)
    xt = torch.tensor(x, requires_grad=True)
    m = torch.tensor(t != -1)
    lt = F.binary_cross_entropy_with_logits(xt[m], torch.tensor(t[t != -1]).float(),
                                            reduction='sum') / max(int(m.sum()), 1)
    lt.backward()
    np.testing.assert_allclose(loss, lt.item(), rtol=1e-5)
    np.testing.assert_allclose(gx, xt.grad.numpy(), rtol=1e-4, atol=1e-7)


def test_softmax_cross_entropy():
    rng = np.random.RandomState(5)
    x = rng.standard_normal((64, 81)).astype(np.float32)
    t = rng.randint(-1, 81, 64).astype(np.int32)
    loss, gx = np_ref.softmax_cross_entropy(x, t)
    xt = torch.tensor(x, requires_grad=True)
    lt = F.cross_entropy(xt, torch.tensor(t).long(), ignore_index=-1)
    lt.backward()
    np.testing.assert_allclose(loss, lt.item(), rtol=1e-5)
    np.testing.assert_allclose(gx, xt.grad.numpy(), rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize('sigma', [1., 3.])
def test_smooth_l1(sigma):
    rng = np.random.RandomState(6)
    pred = rng.standard_normal((200, 4)).astype(np.float32)
    gt = rng.standard_normal((200, 4)).astype(np.float32)
    label = rng.randint(-1, 3, 200).astype(np.int32)
    loss, g = np_ref.fast_rcnn_loc_loss(pred, gt, label, sigma)
    pt = torch.tensor(pred, requires_grad=True)
    beta = 1. / sigma ** 2
    w = torch.tensor((label > 0).astype(np.float32))[:, None]
    lt = F.smooth_l1_loss(pt * w, torch.tensor(gt) * w, reduction='sum', beta=beta) \
        / float((label >= 0).sum())
    lt.backward()
    np.testing.assert_allclose(loss, lt.item(), rtol=1e-5)
    np.testing.assert_allclose(g, pt.grad.numpy(), rtol=1e-4, atol=1e-7)


def test_momentum_sgd():
    rng = np.random.RandomState(7)
    p = rng.standard_normal(100).astype(np.float32)
    g = rng.standard_normal(100).astype(np.float32)
    v = rng.standard_normal(100).astype(np.float32)
    pt = torch.tensor(p.copy(), requires_grad=True)
    opt = torch.optim.SGD([pt], lr=0.01, momentum=0.9, weight_decay=1e-4)
    # torch: buf = m*buf + g'; p -= lr*buf  <=> chainer's v = m*v - lr*g' with v = -lr*buf
    opt.state[pt]['momentum_buffer'] = torch.tensor(-v / 0.01)
    pt.grad = torch.tensor(g)
    opt.step()
    p2, v2 = np_ref.momentum_sgd_wd(p, g, v, 0.01)
    np.testing.assert_allclose(p2, pt.detach().numpy(), rtol=1e-5, atol=1e-6)


def test_loc_loss_pinned_to_reference_functions(golden_dir):
    """tests/golden/loc_loss.npz comes from the reference's own `_fast_rcnn_loc_loss` /
    `_smooth_l1_loss` (models/mask_rcnn_train_chain.py:192-213, oracle/gen_golden.py section 8)."""
    import os
    d = np.load(os.path.join(golden_dir, 'loc_loss.npz'))
    for sigma, key in ((3., 'loss_sigma3'), (1., 'loss_sigma1')):
        loss, _ = np_ref.fast_rcnn_loc_loss(d['pred'], d['gt'], d['label'], sigma)
        np.testing.assert_allclose(loss, float(d[key]), rtol=1e-6)
