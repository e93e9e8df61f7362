"""HIP ROIAlign (through the C ABI) vs the reference's golden vectors and the oracle."""
import os

import numpy as np
import pytest
import torch

import oracle
from chainer_mask_rcnn_amd import functions as F

pytestmark = pytest.mark.gpu

# north_star: within 1e-4 relative for fp32 ROIAlign
RTOL, ATOL = 1e-4, 1e-5


def _run(dev, x, rois, outh, outw, scale, sr, gy, axes='xy'):
    xt = torch.tensor(x, device=dev, requires_grad=True)
    y = F.roi_align_2d(xt, torch.tensor(rois, device=dev), outh, outw, scale, sr, axes=axes)
    y.backward(torch.tensor(gy, device=dev))
    return y.detach().cpu().numpy(), xt.grad.cpu().numpy()


@pytest.mark.parametrize('name', ['roi_align_testgeom_sr0', 'roi_align_testgeom_sr1',
                                  'roi_align_testgeom_sr2', 'roi_align_toy0',
                                  'roi_align_toy1', 'roi_align_toy2'])
def test_reference_golden(dev, golden_dir, name):
    d = np.load(os.path.join(golden_dir, name + '.npz'))
    y, gx = _run(dev, d['x'], d['rois'], int(d['outh']), int(d['outw']),
                 float(d['spatial_scale']), int(d['sampling_ratio']), d['gy'])
    np.testing.assert_allclose(y, d['y'], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(gx, d['gx'], rtol=RTOL, atol=ATOL * 10)


def test_reference_golden_c4like_yx(dev, golden_dir):
    d = np.load(os.path.join(golden_dir, 'roi_align_c4like.npz'))
    y, gx = _run(dev, d['x'], d['rois_yx'], 14, 14, 1. / 16, 0, d['gy'], axes='yx')
    np.testing.assert_allclose(y, d['y'], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(gx, d['gx'], rtol=RTOL, atol=1e-4)


@pytest.mark.parametrize('C', [3, 8, 64, 260])
@pytest.mark.parametrize('sr', [0, 2])
def test_forward_bit_exact_vs_oracle(dev, C, sr):
    rng = np.random.RandomState(C + sr)
    N, H, W = 2, 25, 38
    x = rng.standard_normal((N, C, H, W)).astype(np.float32)
    R = 40
    y1 = rng.uniform(0, H * 16, R); x1 = rng.uniform(0, W * 16, R)
    y2 = np.minimum(y1 + rng.uniform(0, 300, R), H * 16); x2 = np.minimum(x1 + rng.uniform(0, 400, R), W * 16)
    rois = np.stack([rng.randint(0, N, R), x1, y1, x2, y2], 1).astype(np.float32)
    gy = rng.standard_normal((R, C, 7, 7)).astype(np.float32)
    y, gx = _run(dev, x, rois, 7, 7, 1 / 16., sr, gy)
    y_ref = oracle.roi_align_fwd(x, rois, 7, 7, 1 / 16., sr)
    assert np.array_equal(y, y_ref)          # same fp32 op order, no FMA: bit-exact
    gx_ref = oracle.roi_align_bwd(gy, rois, x.shape, 1 / 16., sr)
    np.testing.assert_allclose(gx, gx_ref, rtol=RTOL, atol=1e-4)


@pytest.mark.parametrize('outhw', [(7, 7), (3, 5), (2, 9)])
def test_forward_every_sampling_grid_path_bit_exact(dev, outhw):
    """The forward kernel picks an unrolled body per sampling grid of the RoI ((1,1), (1,2),
    (2,1), (2,2): all taps of 4 / 2 / 2 / 1 bins in flight) and a generic loop otherwise.  RoIs
    built to land on every grid (gh, gw) in {1,2,3}^2 under adaptive sampling, partly outside
    the map (samples the reference skips), output widths that leave a partial last bin group."""
    outh, outw = outhw
    rng = np.random.RandomState(outh * 16 + outw)
    N, C, H, W = 2, 12, 30, 41
    x = rng.standard_normal((N, C, H, W)).astype(np.float32)
    rois = []
    for gh in (1, 2, 3):
        for gw in (1, 2, 3):
            for k in range(4):
                h = (gh - 0.5) * outh * 16.0 + rng.uniform(-3, 3)     # ceil(h / 16 / outh) == gh
                w = (gw - 0.5) * outw * 16.0 + rng.uniform(-3, 3)
                y1 = rng.uniform(-40, H * 16 - 0.5 * h)
                x1 = rng.uniform(-40, W * 16 - 0.5 * w)
                rois.append([rng.randint(0, N), x1, y1, x1 + w, y1 + h])
    rois = np.asarray(rois, np.float32)
    rh = np.maximum((rois[:, 4] - rois[:, 2]) / 16., 1.)
    rw = np.maximum((rois[:, 3] - rois[:, 1]) / 16., 1.)
    grids = set(zip(np.ceil(rh / outh).astype(int).tolist(), np.ceil(rw / outw).astype(int).tolist()))
    assert grids >= {(a, b) for a in (1, 2, 3) for b in (1, 2, 3)}
    y = F.roi_align_2d(torch.tensor(x, device=dev), torch.tensor(rois, device=dev), outh, outw,
                       1 / 16., 0).cpu().numpy()
    assert np.array_equal(y, oracle.roi_align_fwd(x, rois, outh, outw, 1 / 16., 0))
    for sr in (1, 2, 3):
        y = F.roi_align_2d(torch.tensor(x, device=dev), torch.tensor(rois, device=dev), outh, outw,
                           1 / 16., sr).cpu().numpy()
        assert np.array_equal(y, oracle.roi_align_fwd(x, rois, outh, outw, 1 / 16., sr))


def test_processing_order_does_not_change_the_result(dev):
    """roi_align_2d(order=...) only changes which workgroup handles which RoI: forward output and
    gradient are bit-identical for the identity, a random and the spatially sorted order."""
    rng = np.random.RandomState(7)
    N, C, H, W, R = 2, 24, 25, 38, 70
    x = rng.standard_normal((N, C, H, W)).astype(np.float32)
    y1 = rng.uniform(0, H * 16, R); x1 = rng.uniform(0, W * 16, R)
    y2 = np.minimum(y1 + rng.uniform(0, 300, R), H * 16); x2 = np.minimum(x1 + rng.uniform(0, 400, R), W * 16)
    idx = rng.randint(0, N, R)
    rois = np.stack([idx, y1, x1, y2, x2], 1).astype(np.float32)
    gy = torch.tensor(rng.standard_normal((R, C, 7, 7)).astype(np.float32), device=dev)
    orders = [None, rng.permutation(R).astype(np.int32),
              F.roi_spatial_order(rois[:, 1:], idx, 1 / 16.)]
    assert sorted(orders[2].tolist()) == list(range(R))
    outs = []
    for o in orders:
        xt = torch.tensor(x, device=dev, requires_grad=True)
        od = None if o is None else torch.tensor(o, device=dev)
        y = F.roi_align_2d(xt, torch.tensor(rois, device=dev), 14, 14, 1 / 16., axes='yx',
                           bin_stride=2, order=od)
        y.backward(gy)
        outs.append((y.detach().clone(), xt.grad.clone()))
    for y, g in outs[1:]:
        assert torch.equal(y, outs[0][0]) and torch.equal(g, outs[0][1])
    with pytest.raises(TypeError):
        F.roi_align_2d(torch.tensor(x, device=dev), torch.tensor(rois, device=dev), 14, 14, 1 / 16.,
                       axes='yx', order=torch.zeros(R, dtype=torch.int64, device=dev))
    # debug switch: a non-permutation (duplicate / out-of-range entries) is rejected, and the
    # backward entry point refuses a workspace smaller than its own size query
    import importlib
    ra_mod = importlib.import_module('chainer_mask_rcnn_amd.functions.roi_align_2d')
    from chainer_mask_rcnn_amd import _lib
    ra_mod.VALIDATE_ORDER = True
    try:
        bad = torch.tensor(orders[1], device=dev).clone()
        bad[3] = bad[4]
        with pytest.raises(ValueError):
            F.roi_align_2d(torch.tensor(x, device=dev), torch.tensor(rois, device=dev), 14, 14, 1 / 16.,
                           axes='yx', order=bad)
        F.roi_align_2d(torch.tensor(x, device=dev), torch.tensor(rois, device=dev), 14, 14, 1 / 16.,
                       axes='yx', order=torch.tensor(orders[1], device=dev))
    finally:
        ra_mod.VALIDATE_ORDER = False
    need = _lib.load().mrcnn_roi_align_bwd_workspace_bytes(N, H, W, R, 14, 14, 2)
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    gyc = gy.contiguous(memory_format=torch.channels_last)
    gx = torch.empty((N, H, W, C), device=dev)
    args = (_lib.ptr(gyc), _lib.ptr(torch.tensor(rois[:, [0, 2, 1, 4, 3]].copy(), device=dev)), _lib.ptr(gx),
            N, H, W, C, R, 14, 14, 2, 1 / 16., 0, _lib.ptr(ws))
    with pytest.raises(_lib.MrcnnHipError):
        _lib.call('mrcnn_roi_align_bwd_ws', *args, need // 2, _lib.stream_ptr())
    _lib.call('mrcnn_roi_align_bwd_ws', *args, need, _lib.stream_ptr())
    torch.cuda.synchronize()


@pytest.mark.parametrize('sr', [0, 1, 2])
def test_numerical_gradient_on_the_reference_test_geometry(dev, sr):
    """The reference's own backward test (tests/functions_tests/test_roi_align_2d.py:18-41,
    :89-107: chainer.gradient_check.check_backward with atol 5e-4, rtol 5e-3 on a 3x3x12x8
    input, four RoIs incl. a degenerate one, 5x7 bins, scale 0.6): central differences through the
    HIP forward along random directions against the HIP backward (ROIAlign is linear in x)."""
    rng = np.random.RandomState(10 + sr)
    N, C = 3, 3
    x = np.arange(N * C * 12 * 8, dtype=np.float32).reshape((N, C, 12, 8))
    rng.shuffle(x.reshape(-1))
    x = 2 * x / x.size - 1
    rois = np.array([[0, 1, 1, 6, 6], [2, 6, 2, 7, 11], [1, 3, 1, 5, 10], [0, 3, 3, 3, 3]], np.float32)
    gy = rng.uniform(-1, 1, (4, C, 5, 7)).astype(np.float32)
    rd, gyd = torch.tensor(rois, device=dev), torch.tensor(gy, device=dev)

    def fwd(a):
        return F.roi_align_2d(torch.tensor(a, device=dev), rd, outh=5, outw=7, spatial_scale=0.6,
                              sampling_ratio=sr)

    xt = torch.tensor(x, device=dev, requires_grad=True)
    y = F.roi_align_2d(xt, rd, outh=5, outw=7, spatial_scale=0.6, sampling_ratio=sr)
    assert y.dtype == torch.float32 and tuple(y.shape) == gy.shape
    y.backward(gyd)
    gx = xt.grad.double().cpu().numpy()
    eps = 1e-2
    for _ in range(5):
        d = rng.standard_normal(x.shape).astype(np.float32)
        num = ((fwd(x + eps * d).double() - fwd(x - eps * d).double()) * gyd.double()).sum().item() / (2 * eps)
        ana = float((gx * d).sum())
        assert abs(num - ana) <= 5e-4 + 5e-3 * abs(ana), (num, ana)


def test_out_of_range_samples_skipped(dev):
    x = np.ones((1, 4, 4, 4), np.float32)
    rois = np.array([[0, 0, 0, 200, 200], [0, -50, -50, 2, 2]], np.float32)
    gy = np.ones((2, 4, 2, 2), np.float32)
    y, gx = _run(dev, x, rois, 2, 2, 1.0, 2, gy)
    assert np.array_equal(y, oracle.roi_align_fwd(x, rois, 2, 2, 1.0, 2))
    np.testing.assert_allclose(gx, oracle.roi_align_bwd(gy, rois, x.shape, 1.0, 2), rtol=1e-5, atol=1e-6)


def test_empty_rois(dev):
    x = torch.zeros((1, 8, 5, 5), device=dev, requires_grad=True)
    y = F.roi_align_2d(x, torch.zeros((0, 5), device=dev), 7, 7, 1.0)
    assert tuple(y.shape) == (0, 8, 7, 7)


def test_gradient_mass_full_size_property(dev):
    # size-independent property at the BASELINE C2 shape: sum(gx) == sum(gy)
    # (every sample inside the map) and idempotence of a constant map.
    torch.manual_seed(0)
    N, C, H, W, R = 2, 1024, 51, 84, 1024
    g = torch.Generator(device='cpu').manual_seed(1)
    yx = torch.rand((R, 2), generator=g) * torch.tensor([700., 1200.])
    hw = torch.rand((R, 2), generator=g) * torch.tensor([600., 900.]) + 8
    br = torch.minimum(yx + hw, torch.tensor([800., 1333.]))
    rois = torch.cat([torch.randint(0, N, (R, 1), generator=g).float(), yx, br], 1).to(dev)
    x = torch.full((N, C, H, W), 2.5, device=dev).contiguous(memory_format=torch.channels_last)
    x.requires_grad_(True)
    y = F.roi_align_2d(x, rois, 14, 14, 1 / 16., axes='yx')
    assert tuple(y.shape) == (R, C, 14, 14)
    assert torch.allclose(y, torch.full_like(y, 2.5), rtol=1e-5)
    gy = torch.randn((R, 14, 14, C), device=dev).permute(0, 3, 1, 2)
    y.backward(gy)
    s_gx = x.grad.double().sum().item()
    s_gy = gy.double().sum().item()
    assert abs(s_gx - s_gy) <= 1e-4 * gy.double().abs().sum().item()


def test_backward_forms_agree_at_full_c2_size(dev):
    """BASELINE configs[1] shape (2 x 1024 x 51 x 84, 1024 / 1500 RoIs, 14x14 bins read with stride 2):
    the pixel-owner backward (tables + ordered entry lists) against the atomic gather form, on
    benchmark-like small RoIs and on object-sized ones; gradient mass preserved."""
    import importlib
    mod = importlib.import_module('chainer_mask_rcnn_amd.functions.roi_align_2d')
    N, C, H, W = 2, 1024, 51, 84
    g = torch.Generator(device='cpu').manual_seed(3)
    # R = 1500: more RoIs than one scan chunk of the owner kernel (4 per thread x 256 threads)
    for lo, hi, R in ((30., 200., 1024), (32., 600., 1500)):
        yx = torch.rand((R, 2), generator=g) * torch.tensor([760., 1290.])
        hw = torch.rand((R, 2), generator=g) * (hi - lo) + lo
        br = torch.minimum(yx + hw, torch.tensor([800., 1333.]))
        rois = torch.cat([torch.randint(0, N, (R, 1), generator=g).float(), yx, br], 1).to(dev)
        gy = torch.randn((R, 7, 7, C), generator=g).to(dev).permute(0, 3, 1, 2)
        grads = []
        for det in (True, False):
            old = mod.DETERMINISTIC_BACKWARD
            mod.DETERMINISTIC_BACKWARD = det
            try:
                x = torch.zeros((N, C, H, W), device=dev).contiguous(memory_format=torch.channels_last)
                x.requires_grad_(True)
                F.roi_align_2d(x, rois, 14, 14, 1 / 16., axes='yx', bin_stride=2).backward(gy)
                grads.append(x.grad.clone())
            finally:
                mod.DETERMINISTIC_BACKWARD = old
        a, b = grads
        scale = float(b.abs().max())
        assert float((a - b).abs().max()) <= 1e-4 * scale
        s_gx, s_gy = a.double().sum().item(), gy.double().sum().item()
        assert abs(s_gx - s_gy) <= 1e-4 * gy.double().abs().sum().item()


@pytest.mark.parametrize('C', [8, 64])
def test_bin_stride_equals_subsampled_full(dev, C):
    """bin_stride=2 == the even bins of the full 14x14 ROIAlign, forward (bit-exact) and
    backward (the gradient of the subsampled output)."""
    rng = np.random.RandomState(C)
    N, H, W, R = 2, 25, 38, 30
    x = rng.standard_normal((N, C, H, W)).astype(np.float32)
    y1 = rng.uniform(0, H * 16, R); x1 = rng.uniform(0, W * 16, R)
    y2 = np.minimum(y1 + rng.uniform(0, 300, R), H * 16); x2 = np.minimum(x1 + rng.uniform(0, 400, R), W * 16)
    rois = np.stack([rng.randint(0, N, R), x1, y1, x2, y2], 1).astype(np.float32)
    xt = torch.tensor(x, device=dev, requires_grad=True)
    ys = F.roi_align_2d(xt, torch.tensor(rois, device=dev), 14, 14, 1 / 16., bin_stride=2)
    assert tuple(ys.shape) == (R, C, 7, 7)
    y_ref = oracle.roi_align_fwd(x, rois, 14, 14, 1 / 16., 0)
    assert np.array_equal(ys.detach().cpu().numpy(), y_ref[:, :, ::2, ::2])
    gy = rng.standard_normal((R, C, 7, 7)).astype(np.float32)
    ys.backward(torch.tensor(gy, device=dev))
    gy_full = np.zeros((R, C, 14, 14), np.float32)
    gy_full[:, :, ::2, ::2] = gy
    gx_ref = oracle.roi_align_bwd(gy_full, rois, x.shape, 1 / 16., 0)
    np.testing.assert_allclose(xt.grad.cpu().numpy(), gx_ref, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('C,outhw,bs', [(8, (14, 14), 1), (260, (7, 5), 1), (64, (14, 14), 2), (6, (3, 16), 1)])
def test_backward_forms_agree_and_owner_form_is_reproducible(dev, C, outhw, bs):
    """The pixel-owner backward (ordered RoI lists, no atomics) vs the atomic gather form and
    vs the oracle; two runs of the owner form are bit-identical (fixed summation order)."""
    import importlib
    mod = importlib.import_module('chainer_mask_rcnn_amd.functions.roi_align_2d')
    rng = np.random.RandomState(C)
    N, H, W, R = 3, 25, 38, 300
    y1 = rng.uniform(0, H * 16, R); x1 = rng.uniform(0, W * 16, R)
    y2 = np.minimum(y1 + rng.uniform(0, 300, R), H * 16); x2 = np.minimum(x1 + rng.uniform(0, 400, R), W * 16)
    rois = np.stack([rng.randint(0, N, R), x1, y1, x2, y2], 1).astype(np.float32)
    rois[:5, 3:] = rois[:5, 1:3] + 0.3          # sub-pixel RoIs
    rois[5] = [1, 0, 0, W * 16, H * 16]         # whole image
    oh, ow = -(-outhw[0] // bs), -(-outhw[1] // bs)
    gy = rng.standard_normal((R, C, oh, ow)).astype(np.float32)
    x = torch.zeros((N, C, H, W), device=dev)

    def run(deterministic):
        old = mod.DETERMINISTIC_BACKWARD
        mod.DETERMINISTIC_BACKWARD = deterministic
        try:
            xt = x.clone().requires_grad_(True)
            y = F.roi_align_2d(xt, torch.tensor(rois, device=dev), outhw[0], outhw[1], 1 / 16.,
                               bin_stride=bs)
            y.backward(torch.tensor(gy, device=dev))
            return xt.grad.clone()
        finally:
            mod.DETERMINISTIC_BACKWARD = old

    a, b, c = run(True), run(True), run(False)
    assert torch.equal(a, b)
    np.testing.assert_allclose(a.cpu().numpy(), c.cpu().numpy(), rtol=1e-4, atol=1e-4)
    gy_full = np.zeros((R, C) + tuple(outhw), np.float32)
    gy_full[:, :, ::bs, ::bs] = gy
    gx_ref = oracle.roi_align_bwd(gy_full, rois, (N, C, H, W), 1 / 16., 0)
    np.testing.assert_allclose(a.cpu().numpy(), gx_ref, rtol=1e-4, atol=1e-4)
