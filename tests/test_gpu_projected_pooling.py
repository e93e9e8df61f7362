"""Projected pooling (functions/conv.py): res5.a's two 1x1 projections run on the feature map and
ROIAlign pools THEIR outputs — against the reference order, ROIAlign first
(/root/reference/chainer_mask_rcnn/models/mask_rcnn_resnet.py:168-176), which the oracle computes."""
import numpy as np
import pytest
import torch

import oracle
from oracle import np_ref
from chainer_mask_rcnn_amd import _lib
from chainer_mask_rcnn_amd import functions as F
from chainer_mask_rcnn_amd.functions import conv as C
from chainer_mask_rcnn_amd.functions._layout import nhwc, empty_nhwc

pytestmark = pytest.mark.gpu


def _close(got, ref, rtol=1e-4, atol_of_max=1e-5):
    """north_star: 1e-4 relative per element (+ 1e-5 of the tensor's largest magnitude)."""
    ref = np.asarray(ref, np.float64)
    tol = rtol * np.abs(ref) + atol_of_max * np.abs(ref).max()
    bad = np.abs(np.asarray(got, np.float64) - ref) > tol
    assert not bad.any(), 'worst %.3g of tol' % (np.abs(got - ref) / tol).max()


def _rois(rng, R, N, H, W, big=False):
    hi = 600 if big else 200
    y1 = rng.uniform(-8, H * 16 - 16, R); x1 = rng.uniform(-8, W * 16 - 16, R)
    y2 = np.minimum(y1 + rng.uniform(4, hi, R), H * 16 + 6); x2 = np.minimum(x1 + rng.uniform(4, hi, R), W * 16 + 6)
    return np.stack([rng.randint(0, N, R), x1, y1, x2, y2], 1).astype(np.float32)


@pytest.mark.parametrize('Cn', [6, 64, 516, 2048])
@pytest.mark.parametrize('relu', [False, True])
@pytest.mark.parametrize('bs', [1, 2])
def test_affine_epilogue_is_the_plain_kernel_plus_affine(dev, Cn, relu, bs):
    """mrcnn_roi_align_fwd_affine == relu?(mrcnn_roi_align_fwd_ex * scale + shift), bit for bit (one
    multiplication and one addition per element, no contraction), on every sampling-grid body."""
    rng = np.random.RandomState(Cn + bs)
    N, H, W, R = 2, 19, 27, 37
    x = nhwc(torch.tensor(rng.standard_normal((N, Cn, H, W)).astype(np.float32), device=dev))
    rois = torch.tensor(_rois(rng, R, N, H, W, big=True), device=dev)
    scale = torch.tensor(rng.uniform(0.5, 1.5, Cn).astype(np.float32), device=dev)
    shift = torch.tensor(rng.standard_normal(Cn).astype(np.float32), device=dev)
    order = torch.tensor(rng.permutation(R).astype(np.int32), device=dev)
    spec = C.RoiSpec(rois, 14, 14, 1 / 16., bin_stride=bs, order=order)
    got = C._roi_pool_affine(x, spec, scale, shift, relu)
    plain = F.roi_align_2d(x, rois, 14, 14, 1 / 16., bin_stride=bs)
    ref = plain * scale[None, :, None, None] + shift[None, :, None, None]
    if relu:
        ref = torch.relu(ref)
    assert torch.equal(got, ref)
    # and the plain kernel is the C oracle's, bit for bit
    y_ref = oracle.roi_align_fwd(x.cpu().numpy(), rois.cpu().numpy(), 14, 14, 1 / 16., 0)[:, :, ::bs, ::bs]
    assert np.array_equal(plain.cpu().numpy(), y_ref)


def test_affine_entry_point_validates_its_arguments(dev):
    x = empty_nhwc((1, 8, 4, 4), dev)
    rois = torch.zeros((1, 5), device=dev)
    y = empty_nhwc((1, 8, 7, 7), dev)
    with pytest.raises(_lib.MrcnnHipError, match='scale'):
        _lib.call('mrcnn_roi_align_fwd_affine', _lib.ptr(x), _lib.ptr(rois), _lib.ptr(y), 1, 4, 4, 8, 1, 7, 7,
                  1, 1 / 16., 0, None, None, None, 0, _lib.stream_ptr())


@pytest.mark.parametrize('Cn', [512, 2048])
def test_backward_owner_kernel_on_wide_gradients(dev, Cn):
    """The pixel-owner backward with one channel chunk per workgroup (512 and 2048 channels: one
    and two chunks of 256 lanes) against the C oracle's scatter form."""
    rng = np.random.RandomState(Cn)
    N, H, W, R = 2, 13, 21, 24
    rois = _rois(rng, R, N, H, W, big=True)
    gy = rng.standard_normal((R, Cn, 7, 7)).astype(np.float32)
    spec = C.RoiSpec(torch.tensor(rois, device=dev), 14, 14, 1 / 16., bin_stride=2)
    gz = C._roi_pool_bwd(nhwc(torch.tensor(gy, device=dev)), spec, (N, Cn, H, W))
    full = np.zeros((R, Cn, 14, 14), np.float32)
    full[:, :, ::2, ::2] = gy
    ref = oracle.roi_align_bwd(full, rois, (N, Cn, H, W), 1 / 16., 0)
    np.testing.assert_allclose(gz.cpu().numpy(), ref, rtol=1e-4, atol=1e-4)


def _stage(dev, cin, mid, cout, seed):
    from chainer_mask_rcnn_amd.models.resnet_extractor import BuildingBlock
    torch.manual_seed(seed)
    stage = BuildingBlock(3, cin, mid, cout, 2).to(dev)
    with torch.no_grad():
        for name, p in stage.named_parameters():
            if '.bn' in name and name.endswith('.W'):
                p.uniform_(0.5, 1.5)
            elif '.bn' in name:
                p.normal_(0, 0.3)
    return stage


@pytest.mark.parametrize('shape,chans,R,tail', [
    ((2, 17, 23), (64, 32, 128), 40, False),
    ((2, 25, 38), (128, 64, 256), 150, True),      # 128x128 tiles in the pooled layers, fused tail
])
def test_projected_stage_equals_pooling_first(dev, shape, chans, R, tail):
    """building_block(x, roi=spec) against building_block(roi_align_2d(x, bin_stride=2)) — the
    arrangement every earlier round's parity tests pin to the oracle: the forward within 1e-5 of
    the output's scale, the gradient of the map and of every filter within 1e-4 relative L2 per
    tensor (the entrywise statement, which has to name the ReLU decisions, is
    test_projected_block_against_float64_given_the_decisions)."""
    n, h, w = shape
    cin, mid, cout = chans
    stage = _stage(dev, cin, mid, cout, 5)
    rng = np.random.RandomState(R)
    x = torch.randn((n, cin, h, w), device=dev)
    rois = torch.tensor(_rois(rng, R, n, h, w), device=dev)
    spec = C.RoiSpec(rois, 14, 14, 1 / 16., bin_stride=2)
    rows = torch.tensor(np.sort(rng.choice(R, R // 4, replace=False)).astype(np.int64), device=dev) if tail else None
    out = {}
    g = None
    for projected in (False, True):
        for p in stage.parameters():
            p.grad = None
        xt = x.clone().requires_grad_(True)
        if projected:
            y = stage(xt, first_stride=1, roi=spec, tail_rows=rows)
        else:
            pool = F.roi_align_2d(xt, rois, 14, 14, 1 / 16., bin_stride=2)
            y = stage(pool, first_stride=1, tail_rows=rows)
        ys = y if isinstance(y, tuple) else (y,)
        if g is None:
            g = [torch.randn_like(t) for t in ys]
        torch.autograd.backward(ys, g)
        out[projected] = ([t.detach().cpu().numpy() for t in ys], xt.grad.cpu().numpy(),
                          {k: p.grad.cpu().numpy().copy() for k, p in stage.named_parameters() if p.grad is not None})
    (ya, gxa, ga), (yb, gxb, gb) = out[False], out[True]
    for a, b in zip(ya, yb):
        # forward: the two orders differ by fp32 rounding only
        assert np.abs(a - b).max() <= 1e-5 * np.abs(a).max()
    assert set(ga) == set(gb) and 'a.conv4.W' in ga and 'a.conv1.W' in ga
    # gradients: L2 distance per tensor (a ReLU unit within rounding of zero may be decided
    # differently by the two summation orders and moves single entries)
    for k in ga:
        d = np.linalg.norm((ga[k] - gb[k]).ravel()) / np.linalg.norm(ga[k].ravel())
        assert d < 1e-4, (k, d)
    assert np.linalg.norm((gxa - gxb).ravel()) / np.linalg.norm(gxa.ravel()) < 1e-4


def test_projected_block_against_float64_given_the_decisions(dev):
    """One BottleneckA with projected pooling against a float64 torch-CPU graph in the REFERENCE
    order (ROIAlign by the oracle's separable weights, then the block), evaluated with the HIP run's
    ReLU decisions: every entry of every gradient within 1e-4 of its tensor's scale."""
    from chainer_mask_rcnn_amd.models.resnet_extractor import BuildingBlock
    torch.manual_seed(11)
    n, cin, mid, cout, h, w, R = 2, 64, 32, 128, 11, 15, 30
    stage = BuildingBlock(1, cin, mid, cout, 2).to(dev)
    with torch.no_grad():
        for name, p in stage.named_parameters():
            if '.bn' in name and name.endswith('.W'):
                p.uniform_(0.5, 1.5)
            elif '.bn' in name:
                p.normal_(0, 0.3)
    rng = np.random.RandomState(3)
    x = torch.randn((n, cin, h, w), device=dev)
    rois_h = _rois(rng, R, n, h, w)
    rois = torch.tensor(rois_h, device=dev)
    spec = C.RoiSpec(rois, 14, 14, 1 / 16., bin_stride=2)
    C.RELU_TAP = []
    try:
        xt = x.clone().requires_grad_(True)
        y = stage(xt, first_stride=1, roi=spec)
        gy = torch.randn_like(y)
        y.backward(gy)
        (_, (h1, h2, yb)), = C.RELU_TAP
    finally:
        C.RELU_TAP = None
    # the pooling operator as a dense matrix (R*49, N*H*W) in float64, from the C oracle applied to
    # one-hot maps would be (N*H*W) calls; use autograd-free linearity instead: pool every channel
    # of a float64 map with the oracle's float32 weights is not float64 — so build the matrix from
    # the backward oracle (exactly linear): column j = roi_align_bwd of the j-th unit gradient.
    P = np.zeros((R * 49, n * h * w), np.float64)
    eye = np.zeros((R, 1, 14, 14), np.float32)
    for r in range(R):
        for i in range(7):
            for j in range(7):
                eye[:] = 0
                eye[r, 0, 2 * i, 2 * j] = 1
                P[(r * 7 + i) * 7 + j] = oracle.roi_align_bwd(eye, rois_h, (n, 1, h, w), 1 / 16., 0).reshape(-1)
    Pt = torch.tensor(P)
    a = stage.a
    prm = {k: torch.tensor(v.detach().cpu().numpy().astype(np.float64), requires_grad=v.requires_grad)
           for k, v in a.named_parameters()}
    xr = torch.tensor(x.cpu().numpy().astype(np.float64), requires_grad=True)
    pooled = (Pt @ xr.permute(0, 2, 3, 1).reshape(n * h * w, cin)).reshape(R, 7, 7, cin).permute(0, 3, 1, 2)
    conv = torch.nn.functional.conv2d
    aff = lambda t, bn: t * prm[bn + '.W'][None, :, None, None] + prm[bn + '.b'][None, :, None, None]
    m1 = torch.tensor(h1.detach().cpu().numpy() > 0)
    m2 = torch.tensor(h2.detach().cpu().numpy() > 0)
    m3 = torch.tensor(yb.detach().cpu().numpy() > 0)
    p1 = aff(conv(pooled, prm['conv1.W']), 'bn1')
    r1 = p1 * m1
    p2 = aff(conv(r1, prm['conv2.W'], padding=1), 'bn2')
    r2 = p2 * m2
    p3 = aff(conv(r2, prm['conv3.W']), 'bn3') + aff(conv(pooled, prm['conv4.W']), 'bn4')
    yr = p3 * m3
    yr.backward(torch.tensor(gy.cpu().numpy().astype(np.float64)))
    # decisions: a unit decided differently from float64 sits within rounding of zero
    for pre, m in ((p1, m1), (p2, m2), (p3, m3)):
        pre = pre.detach().numpy()
        diff = (pre > 0) != m.numpy()
        assert diff.sum() <= 3 and (np.abs(pre[diff]) <= 1e-5 * np.abs(pre).max()).all()
    _close(y.detach().cpu().numpy(), yr.detach().numpy())
    scale_tol = lambda got, ref: np.abs(got - ref).max() <= 1e-4 * np.abs(ref).max()
    assert scale_tol(xt.grad.cpu().numpy(), xr.grad.numpy())
    for k in ('conv1.W', 'conv2.W', 'conv3.W', 'conv4.W'):
        assert scale_tol(getattr(a, k.split('.')[0]).W.grad.cpu().numpy(), prm[k].grad.numpy()), k


@pytest.mark.parametrize('roi_size', [14, 7], ids=['roi_size 14 (res5 stride 2: every second bin)',
                                                   'roi_size 7 (the class default: res5 stride 1, every bin)'])
def test_head_switch_gives_the_same_predictions(dev, roi_size):
    """ResNetRoIHead with and without projected pooling: class scores, box regressions and mask
    logits agree to 1e-4 of their scale (inference path, no graph)."""
    import chainer_mask_rcnn_amd as cmr
    torch.manual_seed(0)
    head = cmr.models.mask_rcnn_resnet.ResNetRoIHead(50, 5, roi_size, 1 / 16.).to(dev)
    x = torch.randn((2, 1024, 13, 17), device=dev)
    rng = np.random.RandomState(1)
    r5 = _rois(rng, 20, 2, 13, 17)
    rois = torch.tensor(r5[:, [2, 1, 4, 3]], device=dev)        # (y1, x1, y2, x2)
    idx = torch.tensor(r5[:, 0].astype(np.int32), device=dev)
    outs = {}
    with torch.no_grad():
        for projected in (False, True):
            head.projected_pooling = projected
            outs[projected] = [t.cpu().numpy() for t in head(x, rois, idx)]
    for a, b in zip(outs[False], outs[True]):
        assert np.abs(a - b).max() <= 1e-4 * np.abs(a).max()


def test_inference_keeps_the_projected_map_across_head_calls(dev):
    """predict runs the head once per image on ONE batch map: without a graph the two projections of
    the map are computed once (functions.conv.projected_map), reused by the next call on the same
    map, dropped by weights_changed(), and never used when a graph is recorded."""
    import chainer_mask_rcnn_amd as cmr
    torch.manual_seed(0)
    head = cmr.models.mask_rcnn_resnet.ResNetRoIHead(50, 5, 14, 1 / 16.).to(dev)
    x = nhwc(torch.randn((2, 1024, 13, 17), device=dev))      # as the extractor hands it over: channels-last
    rng = np.random.RandomState(1)
    r5 = _rois(rng, 20, 2, 13, 17)
    rois = torch.tensor(r5[:, [2, 1, 4, 3]], device=dev)
    idx = torch.tensor(r5[:, 0].astype(np.int32), device=dev)
    C.weights_changed()
    with torch.no_grad():
        whole = [t.cpu().numpy() for t in head(x, rois, idx)]
        entry = C._proj_cache['entry']
        parts = [head(x, rois[lo:hi], idx[lo:hi]) for lo, hi in ((0, 7), (7, 20))]
        assert C._proj_cache['entry'] is entry            # same map, same filters: reused
    for k in range(3):
        got = np.concatenate([p[k].cpu().numpy() for p in parts], 0)
        # rows of the head are independent (the tile / K-split policy follows the row count: fp32 rounding)
        assert np.abs(got - whole[k]).max() <= 1e-5 * np.abs(whole[k]).max()
    C.weights_changed()
    assert 'entry' not in C._proj_cache
    xg = x.clone().requires_grad_(True)
    out = head(xg, rois, idx)
    assert 'entry' not in C._proj_cache                    # recorded graph: projections inside the node
    out[0].sum().backward()
    assert xg.grad is not None and torch.isfinite(xg.grad).all()


def test_head_with_no_rois_and_training_mode_row_subset(dev):
    """Edge cases of the head's projected path: an empty RoI set takes the reference-order branch
    (nothing to pool), and a recorded graph with a mask-row subset trains (finite gradients for the
    map and for res5.a's two projections)."""
    import chainer_mask_rcnn_amd as cmr
    torch.manual_seed(0)
    head = cmr.models.mask_rcnn_resnet.ResNetRoIHead(50, 5, 14, 1 / 16.).to(dev)
    x = nhwc(torch.randn((1, 1024, 9, 11), device=dev)).requires_grad_(True)
    rng = np.random.RandomState(2)
    r5 = _rois(rng, 12, 1, 9, 11)
    rois = torch.tensor(r5[:, [2, 1, 4, 3]], device=dev)
    idx = torch.tensor(r5[:, 0].astype(np.int32), device=dev)
    rows = torch.tensor([1, 4, 7], dtype=torch.int64, device=dev)
    locs, scores, masks = head(x, rois, idx, mask_rows=rows)
    assert tuple(locs.shape) == (12, 20) and tuple(masks.shape) == (3, 4, 14, 14)
    (locs.sum() + scores.sum() + masks.sum()).backward()
    a = head.res5.a
    for t in (x.grad, a.conv1.W.grad, a.conv4.W.grad):
        assert t is not None and torch.isfinite(t).all() and float(t.abs().sum()) > 0
    with torch.no_grad():
        out = head(x.detach(), rois[:0], idx[:0])
    assert out[0].shape[0] == 0 and out[2].shape[0] == 0
