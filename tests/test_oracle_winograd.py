"""The Winograd algebra of csrc/conv_winograd.h on the CPU: the literal transform constants equal
the Cook-Toom matrices derived (exact rationals) from the interpolation points, and the three
passes reproduce the oracle's direct convolution to float64 round-off."""
import numpy as np
import pytest

from oracle import np_ref, np_winograd as W


def test_literal_constants_equal_the_derived_matrices():
    at, g, bt = W.cook_toom(4, 3)
    np.testing.assert_allclose(at, W.AT, rtol=0, atol=0)
    np.testing.assert_allclose(bt, W.BT, rtol=0, atol=0)       # dyadic: exact
    np.testing.assert_allclose(g, W.G, rtol=1e-15)
    # backward-filter: F(3, 4) on the same points and the SAME B^T
    at2, g2, _ = W.cook_toom(3, 4, bt=W.BT)
    scale = np.array([1, 3, 3, 15, 15, 1.])
    np.testing.assert_allclose(g2 * scale[:, None], W.G4, rtol=1e-14, atol=1e-14)
    np.testing.assert_allclose(at2 / scale[None, :], W.WGRAD_AT, rtol=1e-14, atol=1e-16)
    # dyadic data-side matrices: every entry exactly representable in fp32
    for m in (W.BT, W.AT, W.G4):
        assert np.array_equal(m.astype(np.float32).astype(np.float64), m)


@pytest.mark.parametrize('shape', [(3, 5, 7, 7, 4), (2, 3, 5, 9, 6), (1, 4, 12, 16, 3), (2, 2, 10, 13, 5)])
def test_three_passes_equal_the_direct_convolution(shape):
    N, C, H, Wd, K = shape
    rng = np.random.RandomState(sum(shape))
    x = rng.standard_normal((N, C, H, Wd))
    w = rng.standard_normal((K, C, 3, 3))
    g = rng.standard_normal((N, K, H, Wd))
    y_ref = np_ref.conv2d_fwd(x, w, None, 1, 1)
    gx_ref, gw_ref, _ = np_ref.conv2d_bwd(x, w, g, 1, 1)
    tol = dict(rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(W.conv3x3_fwd(x, w), y_ref, **tol)
    np.testing.assert_allclose(W.conv3x3_dgrad(g, w), gx_ref, **tol)
    np.testing.assert_allclose(W.conv3x3_wgrad(x, g), gw_ref, **tol)


def test_route_selection_for_the_benchmark_shapes(monkeypatch):
    """functions/conv.py:uses_winograd — which layers of the BASELINE configurations take the
    Winograd route (host logic, no GPU): the RoI head's res5 3x3 and the RPN conv1 in both
    workloads, res4 only at the inference batch size, nothing narrower than 256 channels, nothing
    that is not 3x3 / stride 1 / pad 1."""
    from chainer_mask_rcnn_amd.functions import conv as C
    # the SHIPPED thresholds, whatever an earlier test in this process (the GPU fixture lowers the
    # work threshold for its small models) or the environment set
    monkeypatch.setattr(C, 'WINOGRAD_MIN_WORK', 1 << 27)
    monkeypatch.setattr(C, 'WINOGRAD_MIN_CHANNELS', 256)
    monkeypatch.setattr(C, 'USE_WINOGRAD', True)

    def d(N, Cc, H, Wd, K, k=3, s=1, p=1):
        return C.make_desc((N, Cc, H, Wd), (K, Cc, k, k), s, p)
    assert C.uses_winograd(d(1024, 512, 7, 7, 512))          # res5, train (1024 sampled RoIs)
    assert C.uses_winograd(d(1000, 512, 7, 7, 512))          # res5, inference (per image)
    assert C.uses_winograd(d(2, 1024, 50, 84, 1024))         # RPN conv1, train
    assert C.uses_winograd(d(8, 1024, 64, 64, 1024))         # RPN conv1, inference
    assert C.uses_winograd(d(8, 256, 65, 65, 256))           # res4, inference batch
    assert not C.uses_winograd(d(2, 256, 50, 84, 256))       # res4, train batch: too little work
    assert not C.uses_winograd(d(2, 128, 100, 167, 128))     # res3: narrow
    assert not C.uses_winograd(d(2, 64, 200, 334, 64))       # res2
    assert not C.uses_winograd(d(1024, 512, 7, 7, 512, k=1, p=0))
    assert not C.uses_winograd(d(2, 512, 28, 28, 512, s=2))
