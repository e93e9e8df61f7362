"""The assembled HIP train step against the committed end-to-end fixture
tests/golden/train_step.npz ("the build's CPU restatement run on one tiny
synthetic image with fixed weights -> 6 loss values + selected RoI indices, committed, then
required bit-exact for integers / 1e-4 for floats from the HIP path").  Weights and inputs are
regenerated from the fixture's seeds (oracle/np_step.synthetic_*), the global np.random stream
is seeded as the generator did.

Gradients ENTRY BY ENTRY.  Two fp32-class implementations of this step agree on every entry only if
they take the same ReLU decisions, and among the ~4e7 units with a backward a handful sit within
rounding of zero on ANY seed (tools/pick_step_fixture_seeds.py: 24 of 24 seed pairs have at least one
tensor beyond 1e-4 on BOTH arithmetics) — one flipped decision moves a patch of every gradient below
it by 1e-4 .. 1e-2 of its scale.  The entrywise statement is therefore made in its well-posed form,
for EVERY trainable tensor and with no exemption, on both arithmetics:
  (0) the oracle step run at test time IS the committed artifact: its six losses and every committed
      `grad/*` array agree with the fixture to 1e-6 / 1e-5 of the tensor's scale (BLAS summation
      order on another host; on the generating host they are bit-identical) — the fixture, not the
      test-time run, stays the arbiter;
  (1) the HIP step's ReLU decisions differ from the oracle's only at units whose oracle
      pre-activation lies within 1e-5 of the site's scale of zero (about ten fp32 roundings), at
      most 16 of the ~4e7 units; the flips are printed per site;
  (2) GIVEN the decisions (the oracle step re-evaluated with the HIP step's decisions,
      oracle/np_step.RELU_FORCE), every gradient entry and the six losses agree to 1e-4.
When no decision differs, (2) is the comparison with the committed fixture itself.

Unconditionally (no decision conditioning at all): `test_against_the_fixture_unconditionally` — every
committed entry within 1e-4 for the fp32-MFMA arithmetic in the reference's operation order; 1e-3 (and
>= 99 % of each array within 1e-4) for the default path, whose one flip inside res3 sits above a
committed tensor."""
import os

import numpy as np
import pytest
import torch

import chainer_mask_rcnn_amd as cmr
from oracle import np_step
from oracle.gen_golden import TRAIN_STEP_CFG as C

pytestmark = pytest.mark.gpu


@pytest.fixture(params=['split_bf16x3', 'fp32'])
def arithmetic(request):
    from chainer_mask_rcnn_amd.functions import conv as C_
    C_.set_gemm_arithmetic(request.param)
    yield request.param
    C_.set_gemm_arithmetic(C_.DEFAULT_GEMM_ARITHMETIC)


def _hip_step(dev, with_tap):
    from chainer_mask_rcnn_amd.functions import conv as C_
    P = np_step.synthetic_params(C['n_layers'], seed=C['param_seed'])
    imgs, bboxes, labels, masks, scales = np_step.synthetic_inputs(
        C['input_seed'], C['batch'], C['H'], C['W'], n_gt=C['n_gt'], scale=1.0)
    model = cmr.models.MaskRCNNResNet(
        C['n_layers'], n_fg_class=80, anchor_scales=(2, 4, 8, 16, 32), roi_size=14,
        min_size=C['H'], max_size=C['W'], proposal_creator_params=C['proposal_creator_params'])
    chain = cmr.models.MaskRCNNTrainChain(
        model, proposal_target_creator=cmr.models.utils.ProposalTargetCreator(n_sample=C['n_sample']))
    chain.mask_branch_fg_only = False          # the reference evaluates the mask head on all RoIs
    chain.to(dev).train()
    with torch.no_grad():
        for name, p in model.named_parameters():
            assert tuple(p.shape) == P[name].shape, name
            p.copy_(torch.from_numpy(P[name]))
    np.random.seed(C['np_random_seed'])
    tap = [] if with_tap else None
    C_.RELU_TAP = tap
    try:
        loss = chain(torch.tensor(imgs, device=dev), bboxes, labels, masks, list(scales))
    finally:
        C_.RELU_TAP = None
    after = int(np.random.randint(0, 2 ** 31 - 1))       # position of the global stream
    loss.backward()
    torch.cuda.synchronize()
    return model, chain, tap, after, (P, imgs, bboxes, labels, masks, scales)


def _hip_decisions(tap):
    """RELU_TAP entries in call order -> {oracle site: bool NCHW array} for the sites with a backward."""
    names = []
    for stage, n in zip(('extractor.res2', 'extractor.res3', 'extractor.res4'), np_step.BLOCKS[C['n_layers']]):
        names += ['%s.%s' % (stage, 'a' if i == 0 else 'b%d' % i) for i in range(n)]
    names += ['rpn.conv1'] + ['head.res5.%s' % b for b in ('a', 'b1', 'b2')] + ['head.deconv6']
    assert len(tap) == len(names), (len(tap), len(names))
    out = {}
    for name, (kind, t) in zip(names, tap):
        if name.startswith('extractor.res2'):
            continue                                       # below freeze_at: no backward
        if kind == 'block':
            for sub, a in zip(('1', '2', '3'), t):
                out['%s.%s' % (name, sub)] = (a > 0).cpu().numpy()
        else:
            out[name] = (t > 0).cpu().numpy()
    return out


DECISION_WINDOW = 1e-5      # |oracle pre-activation| / site scale below which a decision may differ
MAX_FLIPS = 16              # absolute cap over all ~4e7 units with a backward


def _assert_oracle_is_the_fixture(free, d):
    """The oracle step run at test time reproduces the committed artifact (losses 1e-6 relative,
    committed gradient arrays 1e-5 of their scale: BLAS summation order on another host)."""
    for k, v in zip(d['loss_names'], d['loss_values']):
        got = float(free['losses'][str(k)])
        assert abs(got - v) <= 1e-6 * max(abs(v), 1e-3), ('oracle vs fixture', k, got, v)
    n = 0
    for key in d.files:
        if key.startswith('grad/'):
            scale = np.abs(d[key]).max()
            assert np.abs(free['grads'][key[5:]] - d[key]).max() <= 1e-5 * scale, ('oracle vs fixture', key)
            n += 1
    assert n > 0
    for k, l2 in zip(d['grad_names'], d['grad_l2']):
        g = np.sqrt(np.sum(free['grads'][str(k)].astype(np.float64) ** 2))
        assert abs(g - l2) <= 1e-5 * l2 + 1e-12, ('oracle vs fixture', k, g, l2)


def _compare_decisions(hip, pre):
    n_units = n_diff = 0
    worst, flips = 0., {}
    for site, m in hip.items():
        p_ = pre[site]
        assert m.shape == p_.shape, (site, m.shape, p_.shape)
        diff = m != (p_ > 0)
        n_units += m.size
        if diff.any():
            n_diff += int(diff.sum())
            flips[site] = int(diff.sum())
            w = float(np.abs(p_[diff]).max() / np.abs(p_).max())
            worst = max(worst, w)
            assert w <= DECISION_WINDOW, (site, w)
    return n_units, n_diff, worst, flips


_ORACLE_FREE = {}


def _oracle_step(inputs, force=None, record=None):
    P, imgs, bboxes, labels, masks, scales = inputs
    np.random.seed(C['np_random_seed'])
    np_step.RELU_FORCE, np_step.RELU_PRE = force, record
    try:
        return np_step.train_step(P, imgs, bboxes, labels, masks, scales, n_layers=C['n_layers'],
                                  n_sample=C['n_sample'],
                                  proposal_creator_params=C['proposal_creator_params'])
    finally:
        np_step.RELU_FORCE = np_step.RELU_PRE = None


def test_hip_train_step_matches_fixture(dev, golden_dir, arithmetic):
    d = np.load(os.path.join(golden_dir, 'train_step.npz'))
    model, chain, tap, rng_after, inputs = _hip_step(dev, with_tap=True)
    # integers: the proposals (decode, top-k order, NMS keep list), the sampled RoIs and their
    # labels, and the position of the global np.random stream afterwards
    # (the box COORDINATES are fp32 results of the RPN convolutions: within 1e-4 relative)
    t = chain.last_targets
    assert np.array_equal(t['gt_roi_labels'].cpu().numpy(), d['gt_roi_labels'])
    assert np.array_equal(t['gt_roi_masks'].cpu().numpy(), d['gt_roi_masks'].astype(np.int32))
    assert np.array_equal(t['gt_rpn_labels'].cpu().numpy(), d['gt_rpn_labels'].astype(np.int32))
    assert rng_after == int(d['np_random_after'])
    np.testing.assert_allclose(t['sample_rois'].cpu().numpy(), d['sample_rois'], rtol=1e-4, atol=1e-3)
    # floats: the six reported scalars
    rep = {k: float(v) for k, v in chain.report.items()}
    for k, v in zip(d['loss_names'], d['loss_values']):
        assert abs(rep[str(k)] - v) <= 1e-4 * max(abs(v), 1e-3), (k, rep[str(k)], v)
    # gradients: the L2 norm of every trainable tensor against the fixture
    grads = {n: p.grad.detach().cpu().numpy() for n, p in model.named_parameters() if p.grad is not None}
    for k, l2, mx in zip(d['grad_names'], d['grad_l2'], d['grad_absmax']):
        g = np.sqrt(np.sum(grads[str(k)].astype(np.float64) ** 2))
        assert abs(g - l2) <= 1e-4 * l2 + 1e-12, (k, g, l2)

    # (1) ReLU decisions against the oracle's own (one free oracle run, shared by both arithmetics)
    hip = _hip_decisions(tap)
    del tap[:]
    if 'out' not in _ORACLE_FREE:
        pre = {}
        _ORACLE_FREE['out'] = _oracle_step(inputs, record=pre)
        _ORACLE_FREE['pre'] = pre
    free, pre = _ORACLE_FREE['out'], _ORACLE_FREE['pre']
    _assert_oracle_is_the_fixture(free, d)
    assert sorted(hip) == sorted(pre), (sorted(set(hip) ^ set(pre)))
    n_units, n_diff, worst, flips = _compare_decisions(hip, pre)
    print('%s: %d of %d ReLU decisions differ from the oracle\'s, all within %.1e of the site scale of zero; '
          'per site: %s' % (arithmetic, n_diff, n_units, worst, flips))
    assert n_diff <= MAX_FLIPS, flips

    # (2) every gradient entry, given the decisions
    if n_diff == 0:
        ref, what = free, 'the fixture step'
        for key in d.files:                 # the committed arrays themselves
            if key.startswith('grad/'):
                scale = np.abs(d[key]).max()
                assert np.abs(grads[key[5:]] - d[key]).max() <= 1e-4 * scale, key
    else:
        ref, what = _oracle_step(inputs, force=hip), 'the oracle step given the HIP decisions'
        for k, v in ref['losses'].items():
            assert abs(rep[k] - v) <= 1e-4 * max(abs(v), 1e-3), (k, rep[k], v)
    worst, worst_name = 0., None
    for name, g_ref in ref['grads'].items():
        err = float(np.abs(grads[name] - g_ref).max() / max(np.abs(g_ref).max(), 1e-30))
        if err > worst:
            worst, worst_name = err, name
    print('%s: worst gradient entry vs %s: %.2e of the tensor scale (%s), %d tensors'
          % (arithmetic, what, worst, worst_name, len(ref['grads'])))
    assert worst <= 1e-4, (worst_name, worst)


@pytest.mark.parametrize('arith,projected,bound', [('fp32', False, 1e-4), ('split_bf16x3', True, 1e-3)],
                         ids=['fp32 / reference order', 'default path'])
def test_against_the_fixture_unconditionally(dev, golden_dir, arith, projected, bound):
    """No decision conditioning at all: the HIP step against the COMMITTED gradient arrays, entry by
    entry.  No configuration of this step takes every ReLU decision of the fixture (about ten of 3.9e7
    units sit within 3e-7 of their site's scale of zero on every arithmetic), so what holds
    unconditionally depends on where the flipped units sit:
      * fp32-MFMA arithmetic in the reference's operation order: its flips are all in the RoI head,
        above every committed tensor's own ReLUs — EVERY committed entry within 1e-4 of its tensor's
        scale (measured 8.4e-6);
      * the default path (split operands, projected pooling): one flip inside res3 moves 0.74 % of the
        entries of extractor.res3.a.conv1.W by up to 2.7e-4 of its scale — bound 1e-3, at least 99 %
        of every committed array within 1e-4.  (Given the decisions: 1.9e-6, the test above.)"""
    from chainer_mask_rcnn_amd.functions import conv as C_
    d = np.load(os.path.join(golden_dir, 'train_step.npz'))
    saved = C_.PROJECTED_POOLING
    C_.set_gemm_arithmetic(arith)
    C_.PROJECTED_POOLING = projected
    try:
        model, chain, tap, rng_after, inputs = _hip_step(dev, with_tap=True)
    finally:
        C_.PROJECTED_POOLING = saved
        C_.set_gemm_arithmetic(C_.DEFAULT_GEMM_ARITHMETIC)
    hip = _hip_decisions(tap)
    del tap[:]
    if 'out' not in _ORACLE_FREE:
        pre = {}
        _ORACLE_FREE['out'] = _oracle_step(inputs, record=pre)
        _ORACLE_FREE['pre'] = pre
    _assert_oracle_is_the_fixture(_ORACLE_FREE['out'], d)
    n_units, n_diff, worst_pre, flips = _compare_decisions(hip, _ORACLE_FREE['pre'])
    assert n_diff <= MAX_FLIPS, flips
    grads = {n: p.grad.detach().cpu().numpy() for n, p in model.named_parameters() if p.grad is not None}
    n, worst, worst_frac = 0, 0., 0.
    for key in d.files:
        if key.startswith('grad/'):
            scale = np.abs(d[key]).max()
            err = np.abs(grads[key[5:]] - d[key]) / scale
            worst = max(worst, float(err.max()))
            worst_frac = max(worst_frac, float((err > 1e-4).mean()))
            n += 1
    print('%s, projected pooling %s, unconditional: %d flips %s; worst committed entry %.2e of its scale, '
          'largest fraction of a tensor beyond 1e-4: %.2e' % (arith, projected, n_diff, flips, worst, worst_frac))
    assert n > 0 and rng_after == int(d['np_random_after'])
    assert worst <= bound and worst_frac <= 1e-2, (worst, worst_frac)
