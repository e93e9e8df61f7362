"""End-to-end: the assembled HIP model (extractor, RPN, head, losses, backward, SGD) vs a
plain PyTorch float64 CPU reference of the same graph with the same weights and the same
targets, on a tiny image — for ResNet-50 (BASELINE configs[1]) and ResNet-101 (configs[3]).
Integer outputs (proposals) are compared with the oracle."""
import numpy as np
import pytest
import torch

import chainer_mask_rcnn_amd as cmr
from chainer_mask_rcnn_amd import optimizers
from oracle import np_ref
import ref_model

pytestmark = pytest.mark.gpu

H, W = 160, 224


def _rel(got, ref):
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    return ((got - ref).abs().max() / ref.abs().max().clamp_min(1e-12)).item()


@pytest.fixture(scope='module', params=[50, 101])
def setup(dev, request):
    return _build(dev, request.param)


_BLOCKS = {50: (3, 4, 6), 101: (3, 4, 23)}      # models/resnet_extractor.py:93-124


def _build(dev, n_layers=50):
    torch.manual_seed(0)
    np.random.seed(0)
    model = cmr.models.MaskRCNNResNet(
        n_layers, n_fg_class=80, anchor_scales=(2, 4, 8, 16, 32), roi_size=14, min_size=H, max_size=W,
        proposal_creator_params=dict(min_size=0, n_train_pre_nms=600, n_train_post_nms=100,
                                     n_test_pre_nms=300, n_test_post_nms=50))
    # make the affine layers non-trivial
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, cmr.links.AffineChannel2D):
                m.W.uniform_(0.4, 0.9)
                m.b.normal_(0, 0.1)
        if n_layers == 101:
            # 23 residual blocks with random filters: damp each block's residual branch so the
            # activations stay O(1) through res4 (trained BN-derived affines do this)
            for name, m in model.extractor.res4.named_modules():
                if name.endswith('bn3'):
                    m.W.mul_(0.5)
        model.rpn.conv1.b.normal_(0, 0.1)
        model.head.deconv6.b.normal_(0, 0.1)
    chain = cmr.models.MaskRCNNTrainChain(
        model, proposal_target_creator=cmr.models.utils.ProposalTargetCreator(n_sample=32))
    chain.to(dev).train()
    rng = np.random.RandomState(0)
    imgs = rng.uniform(-120, 130, (2, 3, H, W)).astype(np.float32)
    bboxes = [np.array([[20, 30, 120, 160], [60, 100, 150, 210]], np.float32),
              np.array([[10, 10, 90, 80]], np.float32)]
    labels = [np.array([3, 17], np.int32), np.array([42], np.int32)]
    masks = []
    for b in bboxes:
        m = np.zeros((len(b), H, W), np.int32)
        for g, (y0, x0, y1, x1) in enumerate(b.astype(int)):
            m[g, y0 + 5:y1 - 5, x0 + 5:x1 - 5] = 1
        masks.append(m)
    return model, chain, imgs, bboxes, labels, masks


def freeze_like_reference(model, chain):
    """examples/train_common.py:185-190: conv1, bn1, res2 and every AffineChannel2D."""
    optimizers.disable_update(model.extractor.conv1)
    optimizers.disable_update(model.extractor.bn1)
    optimizers.disable_update(model.extractor.res2)
    for m in chain.modules():
        if isinstance(m, cmr.links.AffineChannel2D):
            optimizers.disable_update(m)


def test_extractor_and_rpn_forward(dev, setup):
    model, chain, imgs, *_ = setup
    blocks = _BLOCKS[len(model.extractor.res4._names) == 23 and 101 or 50]
    P = ref_model.RefParams(model)
    x = torch.tensor(imgs)
    with torch.no_grad():
        feat = model.extractor(torch.tensor(imgs, device=dev))
        feat_ref = ref_model.extractor(x, P, blocks=blocks)
        assert tuple(feat.shape) == tuple(feat_ref.shape) == (2, 1024, 11, 15)
        assert _rel(feat, feat_ref) < 1e-4
        model.eval()
        locs, scores, rois, roi_indices, anchor = model.rpn(feat, (H, W), [1., 1.])
        model.train()
        locs_ref, scores_ref = ref_model.rpn(feat_ref, P, 15)
        assert _rel(locs, locs_ref) < 1e-4 and _rel(scores, scores_ref) < 1e-4
    # proposals: same integer decisions as the oracle's ProposalCreator on the HIP outputs
    pc = np_ref.ProposalCreator(min_size=0, n_test_pre_nms=300, n_test_post_nms=50)
    for i in range(2):
        ref_roi = pc(locs[i].cpu().numpy(), scores[i].cpu().numpy(), anchor.cpu().numpy(),
                     (H, W), 1., train=False)
        got = rois[(roi_indices == i)].cpu().numpy()
        assert got.shape == ref_roi.shape and np.array_equal(got, ref_roi)


def test_train_step_matches_reference(dev, setup, monkeypatch):
    """The round-1/2 entrywise criterion, kept on the route it was written for (the direct head
    forward): it is decided by which units sit within rounding of zero (profiles/r03_seed_study.json)
    and therefore says little about any fp32 implementation; the well-posed statement for BOTH
    routes is test_train_step_gradients_given_the_relu_decisions below."""
    from chainer_mask_rcnn_amd.functions import conv as C
    monkeypatch.setattr(C, 'WINOGRAD_TRAIN_FORWARD', 'conv2d')
    model, chain, imgs, bboxes, labels, masks = setup
    for p in chain.parameters():
        p.grad = None
    np.random.seed(123)
    loss = chain(torch.tensor(imgs, device=dev), bboxes, labels, masks, [1., 1.])
    loss.backward()
    rep = {k: float(v) for k, v in chain.report.items()}

    # CPU reference with the same sampled RoIs / targets (replay the RNG stream)
    blocks = _BLOCKS[len(model.extractor.res4._names) == 23 and 101 or 50]
    P = ref_model.RefParams(model)
    feat = ref_model.extractor(torch.tensor(imgs), P, blocks=blocks)
    rl, rs = ref_model.rpn(feat, P, 15)
    with torch.no_grad():
        locs, scores, rois, roi_indices, anchor = model.rpn(
            model.extractor(torch.tensor(imgs, device=dev)), (H, W), [1., 1.])
    np.random.seed(123)
    ptc = chain.proposal_target_creator
    s_rois, s_idx, g_locs, g_labels, g_masks = [], [], [], [], []
    rois_h, idx_h = rois.cpu().numpy(), roi_indices.cpu().numpy()
    for i in range(2):
        a, b, c, d = ptc(rois_h[idx_h == i], bboxes[i], labels[i], masks[i])
        s_rois.append(a); s_idx.append(np.full(len(a), i, np.int32))
        g_locs.append(b); g_labels.append(c); g_masks.append(d)
    atc = chain.anchor_target_creator
    r_locs, r_labels = zip(*[atc(b, anchor.cpu().numpy(), (H, W)) for b in bboxes])
    cat = lambda xs, dt: torch.tensor(np.concatenate(xs, 0), dtype=dt)
    cls_locs, sc, mk = ref_model.head(feat, cat(s_rois, torch.float32), cat(s_idx, torch.int32),
                                      P, 81, 14)
    parts = ref_model.losses(rl, rs, cat(r_locs, torch.float32), cat(r_labels, torch.int32),
                             cls_locs, sc, mk, cat(g_locs, torch.float32),
                             cat(g_labels, torch.int32), cat(g_masks, torch.int32))
    ref_loss = sum(parts)
    ref_loss.backward()
    names = ['rpn_loc_loss', 'rpn_cls_loss', 'roi_loc_loss', 'roi_cls_loss', 'roi_mask_loss']
    for n, v in zip(names, parts):
        assert abs(rep[n] - v.item()) <= 1e-4 * max(abs(v.item()), 1e-3), (n, rep[n], v.item())
    assert abs(rep['loss'] - ref_loss.item()) <= 1e-4 * abs(ref_loss.item())

    # gradients of every trainable parameter, against the float64 graph: north_star's 1e-4
    # relative for fp32 conv, measured against the gradient tensor's scale.
    # A ReLU whose pre-activation lies within fp32 rounding of zero is decided differently by
    # ANY two fp32 summation orders (and by the float64 graph): one such flip moves one row of
    # the adjacent weight gradients by the contribution of one pixel, ~1e-4..1e-3 of the
    # tensor's scale.  tools/grad_floor.py measures this floor with torch's CPU fp32 kernels on
    # the same graph: worst layer 5.8e-4 (R-50) / 3.1e-4 (R-101) for CPU-fp32, 6.1e-5 / 5.6e-4
    # for the HIP path (different layers flip in each).  Hence: every tensor has >= 99.9 % of
    # its entries within 1e-4, and no entry is off by more than 2e-3.
    worst, worst_name, worst_frac = 0., None, 0.
    for name, p in model.named_parameters():
        g_ref = P['' + name].grad
        if name.startswith('extractor.conv1') or name.startswith('extractor.bn1') \
                or name.startswith('extractor.res2') or '.bn' in name:
            continue
        assert p.grad is not None, name
        assert g_ref is not None, name
        got, ref = p.grad.detach().cpu().double(), g_ref.detach().double()
        err = (got - ref).abs() / ref.abs().max().clamp_min(1e-12)
        frac = float((err > 1e-4).double().mean())
        worst_frac = max(worst_frac, frac)
        assert frac <= 1e-3, (name, frac)
        if float(err.max()) > worst:
            worst, worst_name = float(err.max()), name
    print('worst relative gradient error %.3e (%s), largest fraction of entries beyond 1e-4: %.2e'
          % (worst, worst_name, worst_frac))
    assert worst < 2e-3, (worst_name, worst)


def _reference_step(model, chain, dev, imgs, bboxes, labels, masks, blocks, relus):
    """Float64 CPU graph of one train step on the SAME sampled RoIs / targets as the HIP step
    (the samplers' RNG stream is replayed); returns (losses, RefParams with .grad filled)."""
    P = ref_model.RefParams(model)
    feat = ref_model.extractor(torch.tensor(imgs), P, blocks=blocks, relus=relus)
    rl, rs = ref_model.rpn(feat, P, 15, relus=relus)
    with torch.no_grad():
        locs, scores, rois, roi_indices, anchor = model.rpn(
            model.extractor(torch.tensor(imgs, device=dev)), (H, W), [1., 1.])
    np.random.seed(123)
    ptc, atc = chain.proposal_target_creator, chain.anchor_target_creator
    rois_h, idx_h = rois.cpu().numpy(), roi_indices.cpu().numpy()
    parts = [ptc(rois_h[idx_h == i], bboxes[i], labels[i], masks[i]) for i in range(2)]
    r_locs, r_labels = zip(*[atc(b, anchor.cpu().numpy(), (H, W)) for b in bboxes])
    cat = lambda xs, dt: torch.tensor(np.concatenate(xs, 0), dtype=dt)
    s_idx = [np.full(len(p[0]), i, np.int32) for i, p in enumerate(parts)]
    cls_locs, sc, mk = ref_model.head(feat, cat([p[0] for p in parts], torch.float32),
                                      cat(s_idx, torch.int32), P, 81, 14, relus=relus)
    losses = ref_model.losses(rl, rs, cat(r_locs, torch.float32), cat(r_labels, torch.int32),
                              cls_locs, sc, mk, cat([p[1] for p in parts], torch.float32),
                              cat([p[2] for p in parts], torch.int32),
                              cat([p[3] for p in parts], torch.int32))
    sum(losses).backward()
    return losses, P


@pytest.mark.parametrize('route', ['conv2d', True])
def test_train_step_gradients_given_the_relu_decisions(dev, setup, monkeypatch, route):
    """Whole-graph parity in a form that is well-posed at the fp32 floor (DESIGN.md section 4.3,
    profiles/r03_seed_study.json: the entrywise criterion of test_train_step_matches_reference is
    decided by which units happen to sit within rounding of zero, for every fp32 implementation).
    Two statements, for the direct head forward (WINOGRAD_TRAIN_FORWARD = 'conv2d': only the RPN's
    conv1 routed) and the shipped default (True: res5's 3x3 forward on the Winograd route as well):

    (1) DECISIONS.  Every ReLU decision of the HIP step (res3, res4, RPN conv1, res5, deconv6)
        that differs from the float64 graph's belongs to a unit whose float64 pre-activation lies
        within 1e-4 of the tensor's scale of zero (the north star's tolerance for a computed fp32
        value, here including what the layers upstream contributed), and fewer than 1 in 100 000
        units differ (measured: 2-10 of 4e7 direct, 58-67 routed).
    (2) ARITHMETIC.  Given the SAME decisions (the float64 graph evaluated with the HIP step's
        ReLU masks), every entry of every parameter gradient is within 1e-4 of the tensor's scale
        of float64 — no percentile, no exemptions — and the six losses within 1e-4."""
    from chainer_mask_rcnn_amd.functions import conv as C
    model, chain, imgs, bboxes, labels, masks = setup
    blocks = _BLOCKS[len(model.extractor.res4._names) == 23 and 101 or 50]
    monkeypatch.setattr(C, 'WINOGRAD_TRAIN_FORWARD', route)
    monkeypatch.setattr(chain, 'mask_branch_fg_only', False)      # deconv6's ReLU on every row
    tap = []
    monkeypatch.setattr(C, 'RELU_TAP', tap)
    for p in chain.parameters():
        p.grad = None
    np.random.seed(123)
    loss = chain(torch.tensor(imgs, device=dev), bboxes, labels, masks, [1., 1.])
    monkeypatch.setattr(C, 'RELU_TAP', None)
    loss.backward()
    torch.cuda.synchronize()
    rep = {k: float(v) for k, v in chain.report.items()}
    # the taps in call order: res2 (frozen), res3, res4 blocks; RPN conv1; res5 blocks; deconv6
    names = []
    for stage, n in zip(('extractor.res2', 'extractor.res3', 'extractor.res4'), blocks):
        names += ['%s.%s' % (stage, 'a' if i == 0 else 'b%d' % i) for i in range(n)]
    names += ['rpn.conv1'] + ['head.res5.%s' % b for b in ('a', 'b1', 'b2')] + ['head.deconv6']
    assert len(tap) == len(names), (len(tap), len(names))
    hip_masks = {}
    for name, (kind, t) in zip(names, tap):
        if kind == 'block':
            for sub, a in zip(('h1', 'h2', 'y'), t):
                hip_masks['%s.%s' % (name, sub)] = (a > 0).cpu()
        else:
            hip_masks[name] = (t > 0).cpu()
    del tap[:]
    hip_masks = {k: v for k, v in hip_masks.items() if not k.startswith('extractor.res2')}

    # (1) decisions against the float64 graph's own
    rec = ref_model.Relus(record=True)
    _reference_step(model, chain, dev, imgs, bboxes, labels, masks, blocks, rec)
    n_units = n_diff = 0
    worst = 0.
    for name, m in hip_masks.items():
        pre = rec.pre[name]
        assert tuple(pre.shape) == tuple(m.shape), (name, pre.shape, m.shape)
        diff = m != (pre > 0)
        n_units += m.numel()
        n_diff += int(diff.sum())
        if diff.any():
            w = float(pre[diff].abs().max() / pre.abs().max())
            worst = max(worst, w)
            assert w <= 1e-4, (name, w)
    print('%s: %d of %d ReLU decisions differ from float64 (%.1e), all within %.1e of the scale of zero'
          % (route, n_diff, n_units, n_diff / n_units, worst))
    assert n_diff <= 1e-5 * n_units

    # (2) arithmetic, given the decisions
    parts, P = _reference_step(model, chain, dev, imgs, bboxes, labels, masks, blocks,
                               ref_model.Relus(masks=hip_masks))
    for n, v in zip(['rpn_loc_loss', 'rpn_cls_loss', 'roi_loc_loss', 'roi_cls_loss', 'roi_mask_loss'], parts):
        assert abs(rep[n] - v.item()) <= 1e-4 * max(abs(v.item()), 1e-3), (n, rep[n], v.item())
    worst, worst_name = 0., None
    for name, p in model.named_parameters():
        if name.startswith('extractor.conv1') or name.startswith('extractor.bn1') \
                or name.startswith('extractor.res2') or '.bn' in name:
            continue
        ref = P[name].grad.detach().double()
        err = float(((p.grad.detach().cpu().double() - ref).abs() / ref.abs().max().clamp_min(1e-12)).max())
        if err > worst:
            worst, worst_name = err, name
    print('%s: worst gradient entry %.2e of the tensor scale (%s), given the decisions' % (route, worst, worst_name))
    assert worst <= 1e-4, (worst_name, worst)


def test_train_step_with_the_head_forward_on_the_winograd_route(dev, setup, monkeypatch):
    """functions.conv.WINOGRAD_TRAIN_FORWARD = True (the default; bench.py reports the step
    with 'conv2d' as `direct_head_forward`): res5's 3x3 forward convolutions on the F(4x4,3x3)
    route inside a recorded graph.  Its outputs differ from the direct kernel's by ~1e-6 of the tensor scale, which flips
    ten times as many ReLU decisions of units sitting at zero; each flip moves one row of a few
    weight gradients (profiles/r03_seed_study.json: over ten random instances the entrywise
    criterion of test_train_step_matches_reference passes / fails on the SAME instances with and
    without the route).  Held here to: the six losses within 1e-4 of the float64 graph, every
    gradient tensor within 3e-4 of the direct route's gradient in relative L2 norm and no entry
    further than 2e-3 of the tensor's scale from it."""
    from chainer_mask_rcnn_amd.functions import conv as C
    model, chain, imgs, bboxes, labels, masks = setup
    runs = {}
    for mode in ('conv2d', True):
        monkeypatch.setattr(C, 'WINOGRAD_TRAIN_FORWARD', mode)
        for p in chain.parameters():
            p.grad = None
        np.random.seed(123)
        loss = chain(torch.tensor(imgs, device=dev), bboxes, labels, masks, [1., 1.])
        loss.backward()
        torch.cuda.synchronize()
        runs[mode] = ({k: float(v) for k, v in chain.report.items()},
                      {n: p.grad.detach().double().cpu() for n, p in model.named_parameters()
                       if p.grad is not None})
    for k, v in runs['conv2d'][0].items():
        assert abs(runs[True][0][k] - v) <= 1e-4 * max(abs(v), 1e-3), (k, runs[True][0][k], v)
    worst_l2, worst_max, worst_name = 0., 0., None
    for n, g0 in runs['conv2d'][1].items():
        g1 = runs[True][1][n]
        l2 = float((g1 - g0).norm() / g0.norm().clamp_min(1e-30))
        mx = float((g1 - g0).abs().max() / g0.abs().max().clamp_min(1e-30))
        if l2 > worst_l2:
            worst_l2, worst_name = l2, n
        worst_max = max(worst_max, mx)
    print('winograd-forward route vs direct: worst relative L2 %.2e (%s), worst entry %.2e of scale'
          % (worst_l2, worst_name, worst_max))
    assert worst_l2 <= 3e-4 and worst_max <= 2e-3, (worst_name, worst_l2, worst_max)


def test_frozen_prefix_prefetch_is_results_identical(dev):
    """MaskRCNNTrainChain.next_imgs: the next batch's frozen prefix (conv1 .. res2) is queued on a
    side stream when the backbone's backward begins and consumed by the next forward.  Same
    kernels on the same inputs: losses and weights after three steps over alternating batches are
    bit-identical to the plain loop; a batch that was not announced is simply computed."""
    results = []
    for announce in (False, True):
        model, chain, imgs, bboxes, labels, masks = _build(dev, 50)
        opt = optimizers.MomentumSGD(lr=0.002, momentum=0.9)
        opt.setup(chain)
        opt.add_hook(optimizers.WeightDecay(1e-4))
        freeze_like_reference(model, chain)
        xa = torch.tensor(imgs, device=dev)
        xb = torch.tensor(imgs[:, :, :, ::-1].copy(), device=dev)
        seq = [xa, xb, xa, xa]
        np.random.seed(7)
        losses = []
        for k in range(3):
            # step 2 announces a batch that is NOT the one that follows (xb instead of xa)
            chain.next_imgs = (seq[k + 1] if k != 1 else xb) if announce else None
            losses.append(float(opt.update(chain, seq[k], bboxes, labels, masks, [1., 1.]).detach()))
            if announce and k == 0:
                assert model.extractor._prefetched is not None          # queued during backward
        torch.cuda.synchronize()
        results.append((losses, model.extractor.res4.b2.conv2.W.detach().cpu().numpy().copy(),
                        model.head.res5.a.conv1.W.detach().cpu().numpy().copy()))
    assert results[0][0] == results[1][0] and all(np.isfinite(results[0][0]))
    assert np.array_equal(results[0][1], results[1][1]) and np.array_equal(results[0][2], results[1][2])


def test_optimizer_arena_step(dev, setup):
    """MomentumSGD/WeightDecay over the flat arena == per-parameter oracle rule; frozen
    parameters (conv1, bn1, res2, affine) untouched (examples/train_common.py:176-190)."""
    model, chain, imgs, bboxes, labels, masks = setup
    opt = optimizers.MomentumSGD(lr=0.01, momentum=0.9)
    opt.setup(chain)
    opt.add_hook(optimizers.WeightDecay(1e-4))
    freeze_like_reference(model, chain)
    before = {n: p.detach().clone() for n, p in model.named_parameters()}
    np.random.seed(5)
    opt.update(chain, torch.tensor(imgs, device=dev), bboxes, labels, masks, [1., 1.])
    opt.update(chain, torch.tensor(imgs, device=dev), bboxes, labels, masks, [1., 1.])
    torch.cuda.synchronize()
    # update() cleared the gradient arena in the SGD launch (cleargrads of the next iteration)
    assert float(opt.arena.grads.abs().max()) == 0.
    n_train = 0
    for n, p in model.named_parameters():
        if not p.requires_grad:
            assert torch.equal(p, before[n]), n
        else:
            n_train += p.numel()
            assert not torch.equal(p, before[n]), n
            assert torch.isfinite(p).all()
    # trainable parameter count (35.70 M R-50 / 54.64 M R-101 without affine)
    # + fused-head padding
    if len(model.extractor.res4._names) == 23:
        assert 54.5e6 < n_train < 54.8e6
    else:
        assert 35.6e6 < n_train < 35.8e6
    # a third step equals the oracle rule applied to the arena (gradients from a plain
    # forward/backward; step() without zeroing keeps them readable)
    a = opt.arena
    loss = chain(torch.tensor(imgs, device=dev), bboxes, labels, masks, [1., 1.])
    loss.backward()
    from chainer_mask_rcnn_amd.functions.conv import join_wgrad_stream
    join_wgrad_stream()
    assert all(a.written()), 'every arena parameter received a gradient'
    p0, v0 = a.values.clone(), a.momenta.clone()
    g = a.grads.clone()
    opt.step()
    p_ref, v_ref = np_ref.momentum_sgd_wd(p0.cpu().numpy(), g.cpu().numpy(), v0.cpu().numpy(), 0.01)
    np.testing.assert_allclose(a.values.cpu().numpy(), p_ref, rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(a.momenta.cpu().numpy(), v_ref, rtol=1e-5, atol=1e-7)


def test_predict_prepared_runs(dev, setup):
    model, chain, imgs, *_ = setup
    bboxes, roi_masks, labels, scores = model.predict_prepared(
        torch.tensor(imgs, device=dev), [1., 1.], [(H, W), (H, W)])
    assert len(bboxes) == len(roi_masks) == len(labels) == len(scores) == 2
    for b, m, l, s in zip(bboxes, roi_masks, labels, scores):
        assert b.shape[1] == 4 and len(b) == len(m) == len(l) == len(s)
        assert m.shape[1:] == (80, 14, 14)
        assert len(b) <= 100


def test_mask_branch_fg_only_is_results_identical(dev):
    """Running the mask branch on foreground rows only gives the same losses and the same
    parameter gradients as the reference's all-rows evaluation."""
    model, chain, imgs, bboxes, labels, masks = _build(dev)
    out = {}
    for flag in (False, True):
        for p in chain.parameters():
            p.grad = None
        chain.mask_branch_fg_only = flag
        np.random.seed(321)
        loss = chain(torch.tensor(imgs, device=dev), bboxes, labels, masks, [1., 1.])
        loss.backward()
        out[flag] = ({k: float(v) for k, v in chain.report.items()},
                     {n: p.grad.detach().clone() for n, p in model.named_parameters()
                      if p.grad is not None})
    chain.mask_branch_fg_only = True
    for k in out[False][0]:
        assert abs(out[False][0][k] - out[True][0][k]) <= 1e-5 * max(abs(out[False][0][k]), 1e-3), k
    for n, g in out[False][1].items():
        assert _rel(out[True][1][n], g) < 1e-4, n


def test_training_is_bit_reproducible_run_to_run(dev):
    """Same seeds, same inputs -> bit-identical losses and weights after three SGD steps: no
    kernel on the path depends on the order of floating-point atomics (ordered split-K slabs,
    pixel-owner ROIAlign backward, two-stage loss reductions)."""
    def run():
        model, chain, imgs, bboxes, labels, masks = _build(dev)
        opt = optimizers.MomentumSGD(lr=0.002, momentum=0.9)
        opt.setup(chain)
        opt.add_hook(optimizers.WeightDecay(1e-4))
        freeze_like_reference(model, chain)
        x = torch.tensor(imgs, device=dev)
        np.random.seed(5)
        losses = []
        for _ in range(3):
            loss = opt.update(chain, x, bboxes, labels, masks, [1., 1.])
            losses.append(loss.item())
        torch.cuda.synchronize()
        return losses, opt.arena.values.clone()
    l1, w1 = run()
    l2, w2 = run()
    assert all(np.isfinite(l1)) and l1 == l2
    assert torch.equal(w1, w2)


def test_optimizer_ownership_rules(dev):
    """chainer semantics of the flat arena (optimizers.py): an un-disabled AffineChannel2D
    raises; layers below freeze_at are never updated even if left enabled; a parameter without
    a gradient in a step is skipped entirely; a dropped .grad is re-bound; two backward passes
    before one step accumulate."""
    from chainer_mask_rcnn_amd.functions.conv import join_wgrad_stream
    model, chain, imgs, bboxes, labels, masks = _build(dev)
    opt = optimizers.MomentumSGD(lr=1e-4, momentum=0.9)
    opt.setup(chain)
    opt.add_hook(optimizers.WeightDecay(1e-4))
    x = torch.tensor(imgs, device=dev)
    with pytest.raises(ValueError, match='AffineChannel2D'):
        opt.update(chain, x, bboxes, labels, masks, [1., 1.])
    for m in chain.modules():
        if isinstance(m, cmr.links.AffineChannel2D):
            optimizers.disable_update(m)
    # conv1 / res2 deliberately left enabled: unchain_backward -> grad None -> never updated
    before = {n: p.detach().clone() for n, p in model.named_parameters()}
    np.random.seed(5)
    opt.update(chain, x, bboxes, labels, masks, [1., 1.])
    for n, p in model.named_parameters():
        if n.startswith('extractor.conv1') or n.startswith('extractor.res2'):
            assert torch.equal(p, before[n]), n
    arena = opt.arena
    assert all(p is not model.extractor.conv1.W for p in arena.params)

    # (1) no gradient in a step -> the parameter is skipped (no weight decay, no momentum decay):
    # run the head without its mask branch
    w_mask = model.head.mask.W
    i_mask = [i for i, p in enumerate(arena.params) if p is w_mask][0]
    lo, hi = arena.slice_bounds(i_mask, i_mask)
    snap_p, snap_v = arena.values[lo:hi].clone(), arena.momenta[lo:hi].clone()
    assert float(snap_v.abs().max()) > 0
    feat = model.extractor(x)
    t = chain.last_targets
    idx = torch.zeros(len(t['sample_rois']), dtype=torch.int32, device=dev)
    cls_locs, scores, _ = model.head(feat, t['sample_rois'], idx, pred_mask=False)
    (cls_locs.sum() * 1e-4 + scores.sum() * 1e-4).backward()
    join_wgrad_stream()
    written = dict(zip([id(p) for p in arena.params], arena.written()))
    assert not written[id(w_mask)] and not written[id(model.head.deconv6.W)]
    assert written[id(model.head.cls_loc_score.W)] and written[id(model.extractor.res3.a.conv1.W)]
    opt.step(zero_grads=True)
    assert torch.equal(arena.values[lo:hi], snap_p) and torch.equal(arena.momenta[lo:hi], snap_v)
    assert float(arena.grads.abs().max()) == 0.

    # (2) zero_grad(set_to_none=True) drops the arena views: autograd allocates fresh tensors,
    # rebind() (called by step()) folds them in and restores the views
    chain.zero_grad(set_to_none=True)
    np.random.seed(6)
    chain(x, bboxes, labels, masks, [1., 1.]).backward()
    join_wgrad_stream()
    g_foreign = model.head.mask.W.grad.clone()
    assert not arena.aliases(i_mask)
    arena.rebind()
    assert arena.aliases(i_mask) and torch.equal(model.head.mask.W.grad, g_foreign)
    g_once = arena.grads.clone()
    assert torch.isfinite(g_once).all()

    # (3) a second backward before the step accumulates (the first gradient of a step is written
    # in place, later ones go through autograd's accumulation into the same arena view)
    np.random.seed(6)
    chain(x, bboxes, labels, masks, [1., 1.]).backward()
    join_wgrad_stream()
    torch.cuda.synchronize()
    scale = float(g_once.abs().max())
    assert float((arena.grads - 2 * g_once).abs().max()) <= 1e-5 * scale


def test_deferred_weight_gradients_are_results_identical(dev):
    """MomentumSGD.defer_weight_gradients: the weight gradients (and the update) of chosen RoI
    head layers run in the NEXT step's proposal window on a second stream.  Same losses every
    step and, after flush(), bit-identical weights and momenta."""
    def run(defer):
        model, chain, imgs, bboxes, labels, masks = _build(dev)
        opt = optimizers.MomentumSGD(lr=0.002, momentum=0.9)
        opt.setup(chain)
        opt.add_hook(optimizers.WeightDecay(1e-4))
        freeze_like_reference(model, chain)
        if defer:
            a = model.head.res5.a
            opt.defer_weight_gradients([a.conv2.W, a.conv1.W])
        x = torch.tensor(imgs, device=dev)
        np.random.seed(5)
        losses = [opt.update(chain, x, bboxes, labels, masks, [1., 1.]).item() for _ in range(4)]
        pending = opt._pending is not None
        opt.flush()
        torch.cuda.synchronize()
        return losses, opt.arena.values.clone(), opt.arena.momenta.clone(), pending, opt
    l0, w0, v0, p0, _ = run(False)
    l1, w1, v1, p1, opt = run(True)
    assert not p0 and p1                       # work really was held back across the step boundary
    assert l1 == l0
    assert torch.equal(w1, w0) and torch.equal(v1, v0)
    assert float(opt.arena.grads.abs().max()) == 0.    # every slice cleared by its own SGD launch
    # anything that reads parameters outside a train step flushes first
    np.random.seed(5)
    opt.update(opt.target, torch.tensor(_build(dev)[2], device=dev), *_build(dev)[3:], [1., 1.])
    assert opt._pending is not None
    from chainer_mask_rcnn_amd import serializers
    serializers.state_arrays(opt.target.mask_rcnn)
    assert opt._pending is None and opt._join is None
