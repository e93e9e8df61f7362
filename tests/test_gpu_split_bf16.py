"""The two arithmetics of the convolution GEMMs (functions.conv.set_gemm_arithmetic,
csrc/conv_gemm.hip SPLIT).  'split_bf16x3' — the default since round 4 — stages every fp32 operand
element as three bf16 values whose sum is the element exactly and runs six bf16 MFMAs per K step
with fp32 accumulation; 'fp32' is fp32 MFMA everywhere.  Three statements:

(1) ACCURACY.  Against float64 on the same fp32 inputs the split kernels' error is at the fp32 MFMA
    kernel's level (rms within 1.5x, max below 1e-5 of the tensor scale) for the forward form, the
    data gradient and the weight gradient — i.e. this is fp32 arithmetic, not a reduced precision.
(2) PARITY.  The parity tests of the convolution family (tests/test_gpu_conv.py,
    test_gpu_winograd.py: NumPy oracle, north star's 1e-4 per element — they run on the default
    arithmetic in their own files) pass on BOTH arithmetics, on the tile shapes the full-size step
    uses (128x128 forced by the `big_min_tiles` knob, since the test problems are small) and on
    the 64x64 ones.
(3) WHOLE GRAPH.  The well-posed train-step criterion of tests/test_gpu_model.py (ReLU decisions
    against float64 + every gradient entry given the decisions, 1e-4) holds on both."""
import numpy as np
import pytest
import torch

from chainer_mask_rcnn_amd import _lib, functions as F
from chainer_mask_rcnn_amd.functions import conv as C
import test_gpu_conv as TC
import test_gpu_winograd as TW
import test_gpu_model as TM

pytestmark = pytest.mark.gpu


@pytest.fixture(params=['fp32 MFMA, full-size tile policy', 'fp32 MFMA, 128x128 tiles forced',
                        'split operands, 128x128 tiles forced'])
def split(dev, request):
    C.set_gemm_arithmetic('fp32' if request.param.startswith('fp32') else 'split_bf16x3')
    if '128x128' in request.param:
        _lib.set_tuning('big_min_tiles', 1)
    yield request.param
    _lib.set_tuning('big_min_tiles', 384)
    C.set_gemm_arithmetic(C.DEFAULT_GEMM_ARITHMETIC)


def test_switch_rejects_unknown_arithmetic(dev):
    with pytest.raises(ValueError):
        C.set_gemm_arithmetic('bf16')
    assert C.GEMM_ARITHMETIC == C.DEFAULT_GEMM_ARITHMETIC == 'split_bf16x3'


def _errors(got, ref):
    ref = ref.double()
    err = (got.detach().cpu().double() - ref).abs()
    scale = ref.abs().max().item()
    return err.max().item() / scale, err.pow(2).mean().sqrt().item() / scale


@pytest.mark.parametrize('case', [
    # N, C, H, W, K, k: 128x128 tiles (forced), K from 8 to 72 slices, 1x1 and 3x3, ragged edges
    (64, 256, 14, 14, 256, 1),
    (40, 1024, 7, 7, 192, 1),
    (24, 128, 14, 14, 160, 3),
    (2, 256, 51, 84, 256, 3),
])
def test_error_against_float64_is_at_the_fp32_kernels_level(dev, case):
    N, Cc, H, W, K, k = case
    rng = np.random.RandomState(sum(case))
    x = rng.standard_normal((N, Cc, H, W)).astype(np.float32)
    Wt = (rng.standard_normal((K, Cc, k, k)) / np.sqrt(Cc * k * k)).astype(np.float32)
    gy = rng.standard_normal((N, K, H, W)).astype(np.float32)
    xr = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    wr = torch.tensor(Wt, dtype=torch.float64, requires_grad=True)
    yr = torch.nn.functional.conv2d(xr, wr, padding=k // 2)
    yr.backward(torch.tensor(gy, dtype=torch.float64))
    refs = (yr.detach(), xr.grad, wr.grad)
    out = {}
    _lib.set_tuning('big_min_tiles', 1)
    try:
        for kind in ('fp32', 'split_bf16x3'):
            C.set_gemm_arithmetic(kind)
            xt, wt = TC._t(x, dev, True), TC._t(Wt, dev, True)
            y = F.conv2d(xt, wt, None, stride=1, pad=k // 2)
            y.backward(TC._t(gy, dev))
            torch.cuda.synchronize()
            out[kind] = (y.detach(), xt.grad, wt.grad)
    finally:
        C.set_gemm_arithmetic(C.DEFAULT_GEMM_ARITHMETIC)
        _lib.set_tuning('big_min_tiles', 384)
    for what, a, b, ref in zip(('forward', 'data gradient', 'weight gradient'), out['fp32'],
                               out['split_bf16x3'], refs):
        assert not torch.equal(a, b), '%s: the split kernel did not run' % what
        max_f, rms_f = _errors(a, ref)
        max_s, rms_s = _errors(b, ref)
        print('%-16s fp32 MFMA: max %.2e rms %.2e | split: max %.2e rms %.2e (of the tensor scale)'
              % (what, max_f, rms_f, max_s, rms_s))
        assert max_s <= 1e-5, (what, max_s)
        assert rms_s <= 1.5 * rms_f + 1e-9, (what, rms_s, rms_f)


@pytest.mark.parametrize('case', TC.CASES)
def test_conv_parity_cases(dev, split, case):
    TC.test_conv_fwd_dgrad_wgrad(dev, case)


def test_fused_epilogue_and_masked_backward(dev, split):
    TC.test_conv_fused_epilogue_and_backward(dev)


@pytest.mark.parametrize('proj,stride', [(True, 2), (True, 1), (False, 1)])
def test_fused_bottleneck(dev, split, proj, stride):
    TC.test_fused_bottleneck_matches_oracle(dev, proj, stride)


@pytest.mark.parametrize('case', [(2, 256, 51, 84, 256, 3), (2, 1024, 51, 84, 256, 1)])
def test_split_k_tails(dev, split, case):
    for big_split_k in (0, -1):          # 64x64 tiles and the 128x128 tiles cut along K (one-round rule, shipped)
        TC.test_small_m_split_k_leftover_rows(dev, case, big_split_k)


@pytest.mark.parametrize('forward_form', [True, False])
def test_deconv_and_linear(dev, split, forward_form, monkeypatch):
    for dims in ((5, 256, 7, 7, 64), (130, 512, 7, 7, 256)):
        TC.test_deconv2x2s2(dev, dims, forward_form, monkeypatch)
    TC.test_linear(dev)


@pytest.mark.parametrize('case', TW.CASES)
def test_winograd_route(dev, split, case):
    TW.test_winograd_fwd_dgrad_wgrad(dev, case)


def test_train_step_given_the_relu_decisions(dev, split, monkeypatch):
    if '128x128' not in split:
        pytest.skip('the small test model only reaches the 128x128 kernels when they are forced')
    setup = TM._build(dev, 50)
    TM.test_train_step_gradients_given_the_relu_decisions(dev, setup, monkeypatch, True)


# ---- operand ranges the N(0,1) cases do not reach -------------------------------------------------
# Each against float64 PER ELEMENT, 1e-4 |ref| + 1e-5 x (largest |ref| of the element's own output
# ROW): a row is one pixel's outputs, so a scheme that shared an exponent across a tile or a K row
# (block floating point) would fail where magnitudes are mixed; the split is per element and must not.
def _run_1x1(dev, x2d, w2d, kind, force_big):
    """y = x2d (rows, C) @ w2d (K, C)^T through F.conv2d on a (rows, C, 1, 1) tensor."""
    C.set_gemm_arithmetic(kind)
    _lib.set_tuning('big_min_tiles', 1 if force_big else 384)
    try:
        xt = torch.tensor(x2d, device=dev).reshape(x2d.shape[0], x2d.shape[1], 1, 1)
        wt = torch.tensor(w2d, device=dev).reshape(w2d.shape[0], w2d.shape[1], 1, 1)
        y = F.conv2d(xt, wt, None)
        torch.cuda.synchronize()
        return y.reshape(x2d.shape[0], w2d.shape[0]).cpu().numpy()
    finally:
        _lib.set_tuning('big_min_tiles', 384)
        C.set_gemm_arithmetic(C.DEFAULT_GEMM_ARITHMETIC)


def _assert_rowwise(got, ref, what):
    err = np.abs(got.astype(np.float64) - ref)
    bound = 1e-4 * np.abs(ref) + 1e-5 * np.abs(ref).max(axis=1, keepdims=True)
    worst = float((err / bound).max())
    print('%-44s worst error / bound = %.3f' % (what, worst))
    assert worst <= 1.0, (what, worst)
    return err


@pytest.mark.parametrize('force_big', [True, False], ids=['128x128 tiles', '64x64 tiles'])
@pytest.mark.parametrize('mix', ['within every K row', 'row by row', 'column (filter) by column'])
def test_wide_dynamic_range(dev, mix, force_big):
    """Magnitudes from 1e-6 to 1e+6 inside one GEMM: mixed element by element along K, per input
    row, per filter."""
    rng = np.random.RandomState(len(mix))
    M, K, Cc = 640, 192, 512
    x = rng.standard_normal((M, Cc))
    w = rng.standard_normal((K, Cc)) / np.sqrt(Cc)
    if mix == 'within every K row':
        x = x * 10.0 ** rng.uniform(-6, 6, (M, Cc))
    elif mix == 'row by row':
        x = x * 10.0 ** rng.uniform(-6, 6, (M, 1))
    else:
        w = w * 10.0 ** rng.uniform(-6, 6, (K, 1))
    x, w = x.astype(np.float32), w.astype(np.float32)
    ref = x.astype(np.float64) @ w.astype(np.float64).T
    if mix == 'column (filter) by column':          # a column's scale is the filter's: judge columns
        ref_t, tr = ref.T, (lambda a: a.T)
    else:
        ref_t, tr = ref, (lambda a: a)
    e_s = _assert_rowwise(tr(_run_1x1(dev, x, w, 'split_bf16x3', force_big)), ref_t, 'split, ' + mix)
    e_f = _assert_rowwise(tr(_run_1x1(dev, x, w, 'fp32', force_big)), ref_t, 'fp32 MFMA, ' + mix)
    scale = np.abs(ref_t).max(axis=1, keepdims=True)
    rms = lambda e: float(np.sqrt(np.mean((e / scale) ** 2)))
    assert rms(e_s) <= 1.5 * rms(e_f) + 1e-12, (rms(e_s), rms(e_f))


@pytest.mark.parametrize('force_big', [True, False], ids=['128x128 tiles', '64x64 tiles'])
def test_large_common_offset(dev, force_big):
    """Mean-subtracted pixels are O(130) with O(1) structure on top, and a zero-mean filter cancels
    the offset: the result lives 2-3 decimal digits below the summands (3x3 gather path)."""
    rng = np.random.RandomState(7)
    N, Cc, H, W, K = 2, 64, 40, 52, 128
    x = (130.0 + rng.standard_normal((N, Cc, H, W))).astype(np.float32)
    Wt = rng.standard_normal((K, Cc, 3, 3))
    Wt = ((Wt - Wt.mean(axis=(1, 2, 3), keepdims=True)) / np.sqrt(9 * Cc)).astype(np.float32)
    ref = torch.nn.functional.conv2d(torch.tensor(x, dtype=torch.float64), torch.tensor(Wt, dtype=torch.float64),
                                     padding=1).numpy()
    out = {}
    for kind in ('split_bf16x3', 'fp32'):
        C.set_gemm_arithmetic(kind)
        _lib.set_tuning('big_min_tiles', 1 if force_big else 384)
        try:
            out[kind] = F.conv2d(TC._t(x, dev), TC._t(Wt, dev), None, stride=1, pad=1).cpu().numpy()
        finally:
            _lib.set_tuning('big_min_tiles', 384)
            C.set_gemm_arithmetic(C.DEFAULT_GEMM_ARITHMETIC)
    # the summands are 130 x |w| x 576 taps: an fp32 accumulator carries 2^-24 of THAT — the bound
    # is on the summand scale for both arithmetics, and the split must not be worse than fp32 MFMA
    summand = 130.0 * np.abs(Wt.astype(np.float64)).reshape(K, -1).sum(1).max()
    e_s = np.abs(out['split_bf16x3'] - ref).max() / summand
    e_f = np.abs(out['fp32'] - ref).max() / summand
    print('offset 130: max error / summand scale: split %.2e, fp32 MFMA %.2e' % (e_s, e_f))
    assert e_s <= 1e-5 and e_s <= 1.5 * e_f + 1e-9
    # interior outputs (offset fully cancelled, |ref| ~ 1): still 1e-4 relative + floor
    inner = (slice(None), slice(None), slice(1, -1), slice(1, -1))
    err = np.abs(out['split_bf16x3'][inner] - ref[inner])
    assert np.all(err <= 1e-4 * np.abs(ref[inner]) + 1e-5 * summand)


@pytest.mark.parametrize('force_big', [True, False], ids=['128x128 tiles', '64x64 tiles'])
def test_near_flt_max_and_denormal_range(dev, force_big):
    """Up to 0.9 FLT_MAX every element still splits exactly (bf16 shares fp32's exponent range);
    in the denormal range the error is absolute: parts below 2^-126 may be flushed."""
    rng = np.random.RandomState(9)
    M, K, Cc = 256, 128, 256
    fmax = float(np.finfo(np.float32).max)
    # huge: a few elements per row at 0.5 .. 0.9 FLT_MAX, filter small enough that nothing overflows
    x = rng.standard_normal((M, Cc)).astype(np.float32)
    idx = rng.randint(0, Cc, (M, 4))
    for r in range(M):
        x[r, idx[r]] = (rng.uniform(0.5, 0.9, 4) * fmax * rng.choice([-1, 1], 4)).astype(np.float32)
    w = (rng.standard_normal((K, Cc)) * 1e-3 / np.sqrt(Cc)).astype(np.float32)
    ref = x.astype(np.float64) @ w.astype(np.float64).T
    got = _run_1x1(dev, x, w, 'split_bf16x3', force_big)
    assert np.all(np.isfinite(got))
    _assert_rowwise(got, ref, 'split, 0.9 FLT_MAX elements')
    # tiny: magnitudes 1e-44 .. 1e-30 against an O(1) filter
    x = (rng.standard_normal((M, Cc)) * 10.0 ** rng.uniform(-44, -30, (M, Cc))).astype(np.float32)
    w = rng.standard_normal((K, Cc)).astype(np.float32)
    ref = x.astype(np.float64) @ w.astype(np.float64).T
    got = _run_1x1(dev, x, w, 'split_bf16x3', force_big)
    flush = 3 * 2.0 ** -126 * np.abs(w.astype(np.float64)).sum(1)[None, :] + 2.0 ** -126   # per element of x, + the output
    err = np.abs(got.astype(np.float64) - ref)
    bound = 1e-4 * np.abs(ref) + 1e-5 * np.abs(ref).max(axis=1, keepdims=True) + flush
    assert np.all(err <= bound), float((err / bound).max())


@pytest.mark.parametrize('force_big', [True, False], ids=['128x128 tiles', '64x64 tiles'])
def test_non_finite_inputs_behave_as_documented(dev, force_big):
    """functions.conv.set_gemm_arithmetic: a NaN, an infinity or a finite value that rounds to
    infinity in bf16 (> 3.3895e38) makes every output that READS it NaN on the split arithmetic
    (fp32 MFMA: +-inf / NaN as IEEE multiplication gives); outputs that do not read it are
    untouched, bit for bit."""
    rng = np.random.RandomState(11)
    M, K, Cc = 384, 192, 256
    x = rng.standard_normal((M, Cc)).astype(np.float32)
    w = (rng.standard_normal((K, Cc)) / np.sqrt(Cc)).astype(np.float32)
    clean = _run_1x1(dev, x, w, 'split_bf16x3', force_big)
    clean_f = _run_1x1(dev, x, w, 'fp32', force_big)
    bad_rows = {5: np.inf, 130: -np.inf, 131: np.nan, 300: np.float32(3.4e38)}
    xb = x.copy()
    for r, v in bad_rows.items():
        xb[r, 17] = v
    got = _run_1x1(dev, xb, w, 'split_bf16x3', force_big)
    got_f = _run_1x1(dev, xb, w, 'fp32', force_big)
    rows = sorted(bad_rows)
    rest = np.setdiff1d(np.arange(M), rows)
    assert np.array_equal(got[rest], clean[rest]) and np.array_equal(got_f[rest], clean_f[rest])
    assert np.all(np.isnan(got[rows]))                       # the documented split behaviour
    for r in (5, 130):                                       # fp32 MFMA: inf x w = +-inf
        sign = np.sign(xb[r, 17]) * np.sign(w[:, 17])
        assert np.array_equal(got_f[r], np.where(sign > 0, np.inf, -np.inf).astype(np.float32))
    assert np.all(np.isnan(got_f[131]))
    assert np.all(np.isfinite(got_f[300]))                   # 3.4e38 x 0.06 is an ordinary fp32 product
    # the same element in the FILTER operand: every output of that filter's column
    wb = w.copy()
    wb[9, 40] = np.inf
    got = _run_1x1(dev, x, wb, 'split_bf16x3', force_big)
    assert np.all(np.isnan(got[:, 9]))
    assert np.array_equal(np.delete(got, 9, axis=1), np.delete(clean, 9, axis=1))
