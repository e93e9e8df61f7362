"""Winograd F(4x4,3x3) path (csrc/conv_winograd.h, entry points mrcnn_conv3x3_wino_*) against
the NumPy oracle's direct convolution evaluated in float64 on the same fp32 inputs.  Same
per-element criterion as the direct kernels (tests/test_gpu_conv.py): |got - ref| <= 1e-4 |ref|
+ 1e-5 max|ref| — north_star's 1e-4 relative tolerance for fp32 convolutions.  Also pins the
measured error level (max error <= 1e-5 of the tensor scale) so that a regression of the
transform constants or of the interpolation points shows up long before the parity bound."""
import numpy as np
import pytest
import torch

from oracle import np_ref
from chainer_mask_rcnn_amd import _lib
from chainer_mask_rcnn_amd.functions import conv as C
from chainer_mask_rcnn_amd.functions._layout import nhwc
from test_gpu_conv import _close, _64, _t

pytestmark = pytest.mark.gpu

CASES = [
    # N, C, H, W, K
    (256, 128, 7, 7, 128),       # the RoI head's shape class: 2x2 tiles per map, 128x128 GEMM tiles
    (70, 64, 7, 7, 96),          # tile count not a multiple of the GEMM tile
    (33, 36, 5, 9, 40),          # channels not multiples of 32, ragged maps (2x3 tiles)
    (9, 64, 12, 16, 64),         # maps that are whole tiles; 64x64 GEMM tiles
    (3, 32, 10, 13, 48),         # few maps, ragged
]


def _max_rel_to_scale(got, ref):
    ref = np.asarray(ref, np.float64)
    return np.abs(np.asarray(got, np.float64) - ref).max() / max(np.abs(ref).max(), 1e-30)


@pytest.mark.parametrize('case', CASES)
def test_winograd_fwd_dgrad_wgrad(dev, case):
    N, Cc, H, W, K = case
    rng = np.random.RandomState(11)
    x = np.maximum(rng.standard_normal((N, Cc, H, W)), 0).astype(np.float32)
    Wt = (rng.standard_normal((K, Cc, 3, 3)) / np.sqrt(9. * Cc)).astype(np.float32)
    sc = rng.uniform(0.5, 1.5, K).astype(np.float32)
    sh = (0.3 * rng.standard_normal(K)).astype(np.float32)
    xt, wt = nhwc(_t(x, dev)), nhwc(_t(Wt, dev))
    d = C.make_desc(xt.shape, wt.shape, 1, 1)

    # forward with the fused affine + ReLU, transformed input kept
    y, v = C.wino_fwd(xt, wt, d, _t(sc, dev), _t(sh, dev), True, keep_v=True)
    pre = np_ref.conv2d_fwd(*_64(x, Wt), None, 1, 1) * sc[None, :, None, None] + sh[None, :, None, None]
    y_ref = np.maximum(pre, 0)
    _close(y.cpu().numpy(), y_ref)
    assert _max_rel_to_scale(y.cpu().numpy(), y_ref) < 1e-5
    # plain forward (no epilogue, scratch v)
    y0, _ = C.wino_fwd(xt, wt, d, None, None, False)
    _close(y0.cpu().numpy(), np_ref.conv2d_fwd(*_64(x, Wt), None, 1, 1))

    # backward-data: gx = (dgrad(g * rs[k]) * os[c]) masked by (m > 0)
    g = (rng.standard_normal((N, K, H, W)) * (rng.random_sample((N, K, H, W)) < 0.6)).astype(np.float32)
    rs = rng.uniform(0.5, 1.5, K).astype(np.float32)
    os_ = rng.uniform(0.5, 1.5, Cc).astype(np.float32)
    m = rng.standard_normal((N, Cc, H, W)).astype(np.float32)
    gt = nhwc(_t(g, dev))
    gx = C.wino_dgrad(d, gt, wt, fold_scale=_t(rs, dev), out_scale=_t(os_, dev),
                      out_mask_y=nhwc(_t(m, dev)))
    gx_ref, gW_ref, _ = np_ref.conv2d_bwd(*_64(x, Wt, g * rs[None, :, None, None]), 1, 1)
    _close(gx.cpu().numpy(), gx_ref * os_[None, :, None, None] * (m > 0))
    gx1 = C.wino_dgrad(d, gt, wt)
    gx1_ref, gW1_ref, _ = np_ref.conv2d_bwd(*_64(x, Wt, g), 1, 1)
    _close(gx1.cpu().numpy(), gx1_ref)
    assert _max_rel_to_scale(gx1.cpu().numpy(), gx1_ref) < 1e-5

    # backward-filter from the kept transformed input, with and without the row scale
    gW = torch.empty_like(wt)
    C.wino_wgrad_into(d, None, v, gt, gW)
    _close(gW.cpu().numpy(), gW1_ref)
    assert _max_rel_to_scale(gW.cpu().numpy(), gW1_ref) < 1e-5
    C.wino_wgrad_into(d, None, v, gt, gW, row_scale=_t(rs, dev))
    _close(gW.cpu().numpy(), gW_ref)
    # ... and from the raw input (the train step's route: its forward is the direct kernel)
    gW2 = torch.empty_like(wt)
    C.wino_wgrad_into(d, xt, None, gt, gW2, row_scale=_t(rs, dev))
    assert torch.equal(gW2, gW)
    with pytest.raises(_lib.MrcnnHipError):
        C.wino_wgrad_into(d, xt, v, gt, gW2)


def test_winograd_is_deterministic(dev):
    """Ordered slab sums: two runs give bit-identical gradients."""
    N, Cc, H, W, K = 300, 64, 7, 7, 64
    rng = np.random.RandomState(3)
    xt = nhwc(_t(rng.standard_normal((N, Cc, H, W)).astype(np.float32), dev))
    wt = nhwc(_t((rng.standard_normal((K, Cc, 3, 3)) / 24.).astype(np.float32), dev))
    gt = nhwc(_t(rng.standard_normal((N, K, H, W)).astype(np.float32), dev))
    d = C.make_desc(xt.shape, wt.shape, 1, 1)
    outs = []
    for _ in range(2):
        y, v = C.wino_fwd(xt, wt, d, None, None, False, keep_v=True)
        gW = torch.empty_like(wt)
        C.wino_wgrad_into(d, None, v, gt, gW)
        outs.append((y.clone(), C.wino_dgrad(d, gt, wt).clone(), gW.clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_winograd_rejects_other_filters(dev):
    xt = nhwc(torch.zeros((2, 8, 7, 7), device=dev))
    wt = nhwc(torch.zeros((8, 8, 1, 1), device=dev))
    d = C.make_desc(xt.shape, wt.shape, 1, 0)
    with pytest.raises(_lib.MrcnnHipError):
        C.wino_fwd(xt, wt, d, None, None, False)


def test_stage_with_winograd_matches_direct(dev):
    """A res5-like stage over many 7x7 maps: forward and every gradient of the Winograd route
    against the implicit-GEMM route of the same fused stage (both fp32; per-element 1e-4)."""
    from chainer_mask_rcnn_amd.models.resnet_extractor import BuildingBlock
    torch.manual_seed(0)
    stage = BuildingBlock(2, 64, 32, 128, 1).to(dev)
    with torch.no_grad():
        for name, p in stage.named_parameters():
            if '.bn' in name and name.endswith('.W'):
                p.uniform_(0.5, 1.5)
            elif '.bn' in name:
                p.normal_(0, 0.3)
    x = torch.randn(96, 64, 7, 7, device=dev).relu_()
    gy = None
    res = {}
    saved = (C.WINOGRAD_MIN_CHANNELS, C.WINOGRAD_MIN_WORK)
    saved_route = (C.USE_WINOGRAD, C.WINOGRAD_TRAIN_FORWARD)
    C.WINOGRAD_MIN_CHANNELS, C.WINOGRAD_MIN_WORK = 1, 0      # route this narrow stage too
    for use in (False, True):
        C.USE_WINOGRAD = C.WINOGRAD_TRAIN_FORWARD = use
        try:
            xi = x.clone().requires_grad_(True)
            for p in stage.parameters():
                p.grad = None
            y = stage(xi)
            if gy is None:
                gy = torch.randn_like(y)
            y.backward(gy)
            res[use] = [y.detach().cpu().numpy(), xi.grad.cpu().numpy()] + \
                [p.grad.cpu().numpy() for _, p in stage.named_parameters() if p.grad is not None]
        finally:
            C.USE_WINOGRAD, C.WINOGRAD_TRAIN_FORWARD = saved_route
    C.WINOGRAD_MIN_CHANNELS, C.WINOGRAD_MIN_WORK = saved
    assert len(res[True]) == len(res[False]) > 4
    for a, b in zip(res[True], res[False]):
        _close(a, b, rel=1e-4, floor=2e-5)


@pytest.mark.parametrize('train_forward', [False, True])
def test_conv2d_winograd_route_bias_relu(dev, train_forward):
    """F.conv2d on an RPN-conv1-like layer (3x3, bias, ReLU; wide enough to be routed): forward,
    input / filter / bias gradients against the float64 oracle.  In a train step only the
    backward takes the Winograd route (WINOGRAD_TRAIN_FORWARD); under no_grad the forward does."""
    from chainer_mask_rcnn_amd import functions as F
    N, Cc, H, W, K = 2, 256, 19, 30, 256
    rng = np.random.RandomState(5)
    x = np.maximum(rng.standard_normal((N, Cc, H, W)), 0).astype(np.float32)
    Wt = (rng.standard_normal((K, Cc, 3, 3)) / np.sqrt(9. * Cc)).astype(np.float32)
    b = (0.2 * rng.standard_normal(K)).astype(np.float32)
    saved = (C.WINOGRAD_MIN_WORK, C.WINOGRAD_TRAIN_FORWARD)
    C.WINOGRAD_MIN_WORK, C.WINOGRAD_TRAIN_FORWARD = 0, train_forward
    calls = {'fwd': 0, 'dgrad': 0, 'wgrad': 0}
    orig = (C.wino_fwd, C.wino_dgrad, C.wino_wgrad_into)

    def spy(name, fn):
        def wrapped(*a, **k):
            calls[name] += 1
            return fn(*a, **k)
        return wrapped
    C.wino_fwd, C.wino_dgrad, C.wino_wgrad_into = (spy(n, f) for n, f in zip(calls, orig))
    try:
        xt, wt, bt = _t(x, dev, True), _t(Wt, dev, True), _t(b, dev, True)
        y = F.conv2d(xt, wt, bt, stride=1, pad=1, relu=True)
        pre = np_ref.conv2d_fwd(*_64(x, Wt, b), 1, 1)
        _close(y.detach().cpu().numpy(), np.maximum(pre, 0))
        gy = rng.standard_normal(pre.shape).astype(np.float32)
        y.backward(_t(gy, dev))
        g = gy * (y.detach().cpu().numpy() > 0)
        gx, gW, gb = np_ref.conv2d_bwd(*_64(x, Wt, g), 1, 1)
        _close(xt.grad.cpu().numpy(), gx)
        _close(wt.grad.cpu().numpy(), gW)
        _close(bt.grad.cpu().numpy(), gb)
        assert calls == {'fwd': int(train_forward), 'dgrad': 1, 'wgrad': 1}
        with torch.no_grad():
            y2 = F.conv2d(xt, wt, bt, stride=1, pad=1, relu=True)
        assert calls['fwd'] == int(train_forward) + 1
        _close(y2.cpu().numpy(), np.maximum(pre, 0))
    finally:
        C.wino_fwd, C.wino_dgrad, C.wino_wgrad_into = orig
        C.WINOGRAD_MIN_WORK, C.WINOGRAD_TRAIN_FORWARD = saved


def test_inference_filter_cache_follows_weight_updates(dev):
    """Inference calls reuse the transformed filter of an unchanged parameter; an optimizer step
    (which writes the flat arena behind torch's version counters) and an in-place torch write both
    drop it."""
    from chainer_mask_rcnn_amd import functions as F, optimizers
    torch.manual_seed(1)
    saved = C.WINOGRAD_MIN_WORK
    C.WINOGRAD_MIN_WORK = 0
    try:
        conv = torch.nn.Module()
        conv.W = torch.nn.Parameter(torch.randn(256, 256, 3, 3, device=dev) * 0.02)
        x = torch.randn(2, 256, 12, 16, device=dev)

        def infer():
            with torch.no_grad():
                return F.conv2d(x, conv.W, None, stride=1, pad=1)

        def direct():
            C.USE_WINOGRAD = False
            try:
                return infer()
            finally:
                C.USE_WINOGRAD = True
        y0 = infer()
        assert id(conv.W) in C._wino_u_cache
        u0 = C._wino_u_cache[id(conv.W)][2]
        assert infer() is not None and C._wino_u_cache[id(conv.W)][2] is u0     # reused
        _close(y0.cpu().numpy(), direct().cpu().numpy())
        # one SGD step through the arena
        opt = optimizers.MomentumSGD(lr=0.5, momentum=0.)
        opt.setup(conv)
        opt.update(lambda: F.conv2d(x, conv.W, None, stride=1, pad=1).sum() * 1e-3)
        assert id(conv.W) not in C._wino_u_cache
        y1 = infer()
        assert not torch.equal(y1, y0)
        _close(y1.cpu().numpy(), direct().cpu().numpy())
        # an in-place torch write
        with torch.no_grad():
            conv.W.mul_(0.5)
        _close(infer().cpu().numpy(), direct().cpu().numpy())
    finally:
        C.WINOGRAD_MIN_WORK = saved


def test_exact_signs_fixup(dev):
    """MRCNN_EPI_EXACT_SIGNS: the ReLU decisions of the Winograd forward, checked against a float64
    convolution of the same fp32 inputs.  Plain Winograd decides ~3x as many near-zero units
    differently as the direct kernel; with the fix-up (outputs within the propagated rounding bound
    of zero recomputed as direct dot products, a few in 10^5) no more than the direct kernel."""
    import ctypes
    g = torch.Generator(device='cpu').manual_seed(0)
    N, Cc, K, H, W = 384, 512, 512, 7, 7
    x = torch.randn(N, Cc, H, W, generator=g).relu_().mul_(20.)
    w = torch.randn(K, Cc, 3, 3, generator=g) / (3. * Cc ** 0.5)
    sc = torch.rand(K, generator=g) * 0.5 + 0.4
    sh = torch.randn(K, generator=g) * 2.0
    pre = torch.nn.functional.conv2d(x.double(), w.double(), None, 1, 1) \
        * sc.double()[None, :, None, None] + sh.double()[None, :, None, None]
    pre = pre.to(dev)
    scale = pre.abs().max().item()
    xt, wt, sct, sht = nhwc(x.to(dev)), nhwc(w.to(dev)), sc.to(dev), sh.to(dev)
    d = C.make_desc(xt.shape, wt.shape, 1, 1)

    def wrong(y):
        bad = (y > 0) != (pre > 0)
        return int(bad.sum()), (pre.abs()[bad].max().item() / scale if bad.any() else 0.)
    n_direct, _ = wrong(C._fwd_raw(xt, wt, d, sct, sht, None, True))
    y_plain, _ = C.wino_fwd(xt, wt, d, sct, sht, True)
    n_plain, worst_plain = wrong(y_plain)
    y_fix, _ = C.wino_fwd(xt, wt, d, sct, sht, True, exact_signs=True)
    n_fix, worst_fix = wrong(y_fix)
    cnt = ctypes.c_int(0)
    _lib.call('mrcnn_conv3x3_wino_fixup_count', C.ctx_desc(d), _lib.ptr(C._wino_ws(d, dev)),
              _lib.stream_ptr(), ctypes.byref(cnt))
    print('sign disagreements with fp64: direct %d, winograd %d (|pre| up to %.1e of scale), with fix-up %d '
          '(up to %.1e); %d of %d outputs recomputed' % (n_direct, n_plain, worst_plain, n_fix, worst_fix,
                                                         cnt.value, pre.numel()))
    assert n_fix <= max(n_direct, 1) and worst_fix < 2e-7
    assert 0 < cnt.value < 2e-4 * pre.numel()
    # everything else is untouched: same values as the plain route except the recomputed outputs
    changed = int((y_fix != y_plain).sum())
    assert changed <= cnt.value
    _close(y_fix.cpu().numpy(), np.maximum(pre.cpu().numpy(), 0))
