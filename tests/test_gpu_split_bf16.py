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
    TC.test_small_m_split_k_leftover_rows(dev, case)


def test_deconv_and_linear(dev, split):
    TC.test_deconv2x2s2(dev)
    TC.test_linear(dev)


@pytest.mark.parametrize('case', TW.CASES)
def test_winograd_route(dev, split, case):
    TW.test_winograd_fwd_dgrad_wgrad(dev, case)


def test_train_step_given_the_relu_decisions(dev, split, monkeypatch):
    if '128x128' not in split:
        pytest.skip('the small test model only reaches the 128x128 kernels when they are forced')
    setup = TM._build(dev, 50)
    TM.test_train_step_gradients_given_the_relu_decisions(dev, setup, monkeypatch, True)
