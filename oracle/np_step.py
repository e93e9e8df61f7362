"""ORACLE (test infrastructure, not product code) — one whole training iteration on the CPU.

End-to-end restatement of the reference's Chainer CPU path for
`MaskRCNNTrainChain.__call__` + `loss.backward()`
(/root/reference/chainer_mask_rcnn/models/mask_rcnn_train_chain.py:76-189 around
models/resnet_extractor.py:76-90, models/region_proposal_network.py:82-145,
models/mask_rcnn_resnet.py:168-196) composed from the per-layer restatements of this package:
NumPy im2col + BLAS convolutions (what chainer's CPU `Convolution2DFunction` does, SURVEY.md
Appendix A.1), the C ROIAlign / NMS restatements, the chainercv proposal / target creators
(np_ref, np_targets — the global `np.random` stream is consumed in the reference's order) and
the five losses.  The backward pass is written out by hand in reverse layer order (there is no
autograd here); the mask branch runs on EVERY sampled RoI as the reference does (:147-148).

Used as
  * the checker of the assembled HIP train step (tests/golden/train_step.npz is produced by
    this file through oracle/gen_golden.py; tests/test_gpu_step_golden.py),
  * `bench.py`'s `cpu_baseline`: BASELINE configs[0] (one 800x1333 image, 512 RoIs, forward +
    backward) timed in full on the host cores — reported only.

"parity unpinned" below the reference's own files: the layer arithmetic lives in un-vendored
chainer / chainercv (requirements.txt:1-2), see np_ref's header.

Parameters are a dict {name: ndarray} with the product model's / chainer's names and LOGICAL
shapes (conv (out,in,kh,kw), deconv (in,out,2,2), linear (out,in)); `rpn.loc_score` and
`head.cls_loc_score` are the fused (loc rows first, then score rows, zero-padded to a multiple
of 4) filters the product stores — mathematically the reference's two layers side by side.
"""
import time

import numpy as np

from . import np_ref
from . import np_targets

f32 = np.float32
BLOCKS = {50: (3, 4, 6), 101: (3, 4, 23)}       # models/resnet_extractor.py:93-124


# ---------------------------------------------------------------------------------------
# layers with caches
# ---------------------------------------------------------------------------------------
def _aff(x, P, pre):
    return np_ref.affine_channel_2d_fwd(x, P[pre + '.W'], P[pre + '.b'])


# Diagnostic of the fixture generator (oracle/gen_golden.py: choosing the seeds of
# tests/golden/train_step.npz): when set to a dict, every ReLU that has a BACKWARD (res3 .. res5,
# the RPN's conv1, deconv6) records, per site, the smallest |pre-activation| relative to the
# site's largest.  A unit whose pre-activation lies within fp32 rounding of zero takes its
# gradient mask from the rounding; a fixture whose smallest margin is far above that is one on
# which every fp32-class implementation takes the same decisions.  Never changes a result.
RELU_MARGINS = None
# Decision-conditioned evaluation (tests/test_gpu_step_golden.py): RELU_PRE, when a dict, receives
# every such site's pre-activation array; RELU_FORCE, when a dict {site: bool array}, REPLACES the
# site's decisions (p > 0) in the forward AND in the backward's gradient mask — the step "given the
# decisions" of another implementation, which is the well-posed form of an entrywise gradient
# comparison between two fp32-class implementations.  Sites: '<block>.1' / '.2' / '.3' (the three
# ReLUs of a bottleneck), 'rpn.conv1', 'head.deconv6'.  Unset (the default), nothing changes.
RELU_PRE = None
RELU_FORCE = None


def _relu(p, site):
    """-> (relu(p), decisions).  The decisions are (p > 0) unless RELU_FORCE overrides the site."""
    if RELU_MARGINS is not None:
        a = np.abs(p)
        nz = a[a > 0]              # (exact zeros — structural — are zero in every implementation)
        if nz.size:
            RELU_MARGINS[site] = min(RELU_MARGINS.get(site, np.inf), float(nz.min() / a.max()))
    if RELU_PRE is not None:
        RELU_PRE[site] = p
    if RELU_FORCE is not None and site in RELU_FORCE:
        m = np.asarray(RELU_FORCE[site], bool)
        assert m.shape == p.shape, (site, m.shape, p.shape)
        return np.where(m, p, p.dtype.type(0)), m
    y = np.maximum(p, 0)
    return y, y > 0


def bottleneck_fwd(x, P, pre, stride, proj):
    """chainer BottleneckA (proj) / BottleneckB."""
    trainable = not pre.startswith('extractor.res2')          # (res2 is below freeze_at: no backward)
    relu = _relu if trainable else (lambda p, site: (np.maximum(p, 0), None))
    h1, m1 = relu(_aff(np_ref.conv2d_fwd(x, P[pre + '.conv1.W'], None, stride, 0), P, pre + '.bn1'), pre + '.1')
    h2, m2 = relu(_aff(np_ref.conv2d_fwd(h1, P[pre + '.conv2.W'], None, 1, 1), P, pre + '.bn2'), pre + '.2')
    p3 = _aff(np_ref.conv2d_fwd(h2, P[pre + '.conv3.W']), P, pre + '.bn3')
    sc = _aff(np_ref.conv2d_fwd(x, P[pre + '.conv4.W'], None, stride, 0), P, pre + '.bn4') if proj else x
    y, m3 = relu(p3 + sc, pre + '.3')
    return y, (x, h1, h2, y, stride, proj, m1, m2, m3)


def bottleneck_bwd(gy, cache, P, pre, grads, need_gx=True):
    x, h1, h2, y, stride, proj, m1, m2, m3 = cache      # m*: the ReLU decisions ((h > 0) unless forced)
    ch = lambda v: v[None, :, None, None]
    g = gy * m3
    gh2, grads[pre + '.conv3.W'], _ = np_ref.conv2d_bwd(h2, P[pre + '.conv3.W'], g * ch(P[pre + '.bn3.W']))
    g2 = gh2 * m2 * ch(P[pre + '.bn2.W'])
    gh1, grads[pre + '.conv2.W'], _ = np_ref.conv2d_bwd(h1, P[pre + '.conv2.W'], g2, 1, 1)
    g1 = gh1 * m1 * ch(P[pre + '.bn1.W'])
    gx, grads[pre + '.conv1.W'], _ = np_ref.conv2d_bwd(x, P[pre + '.conv1.W'], g1, stride, 0,
                                                        need_gx=need_gx)
    if proj:
        gx4, grads[pre + '.conv4.W'], _ = np_ref.conv2d_bwd(
            x, P[pre + '.conv4.W'], g * ch(P[pre + '.bn4.W']), stride, 0, need_gx=need_gx)
        if need_gx:
            gx = gx + gx4
    elif need_gx:
        gx = gx + g
    return gx


def stage_fwd(x, P, pre, n, stride):
    caches = []
    x, c = bottleneck_fwd(x, P, pre + '.a', stride, True)
    caches.append(c)
    for i in range(1, n):
        x, c = bottleneck_fwd(x, P, pre + '.b%d' % i, 1, False)
        caches.append(c)
    return x, caches


def stage_bwd(gy, caches, P, pre, grads, need_gx=True):
    n = len(caches)
    for i in range(n - 1, 0, -1):
        gy = bottleneck_bwd(gy, caches[i], P, pre + '.b%d' % i, grads)
    return bottleneck_bwd(gy, caches[0], P, pre + '.a', grads, need_gx=need_gx)


# ---------------------------------------------------------------------------------------
# the iteration
# ---------------------------------------------------------------------------------------
def train_step(P, imgs, bboxes, labels, masks, scales, n_layers=50, n_class=81,
               anchor_scales=(2, 4, 8, 16, 32), ratios=(0.5, 1, 2), roi_size=14,
               proposal_creator_params=None, n_sample=512, rpn_sigma=3., roi_sigma=1.,
               backward=True, timings=None):
    """One forward (+ backward) of MaskRCNNTrainChain on host arrays.

    imgs (N,3,H,W) f32 mean-subtracted; bboxes / labels / masks per-image lists; scales (N,).
    Returns dict(losses={6 scalars}, grads={name: ndarray} (trainable parameters), rois=[...],
    sample_rois, gt_roi_labels, ...)."""
    tick = time.perf_counter
    t0 = tick()
    mark = (lambda k: timings.__setitem__(k, tick() - t0)) if timings is not None else (lambda k: None)
    nb = BLOCKS[n_layers]
    N, _, H, W = imgs.shape
    A = len(anchor_scales) * len(ratios)
    # ---- extractor (models/resnet_extractor.py:76-90); conv1..res2 carry no gradient ----
    h = np_ref.conv2d_fwd(imgs, P['extractor.conv1.W'], P['extractor.conv1.b'], 2, 3)
    h = np.maximum(_aff(h, P, 'extractor.bn1'), 0)
    h = np_ref.max_pooling_2d(h)
    h, _ = stage_fwd(h, P, 'extractor.res2', nb[0], 1)
    res2 = h
    res3, c_res3 = stage_fwd(res2, P, 'extractor.res3', nb[1], 2)
    feat, c_res4 = stage_fwd(res3, P, 'extractor.res4', nb[2], 2)
    mark('extractor')
    # ---- RPN (models/region_proposal_network.py:82-145) ----
    hh, ww = feat.shape[2:]
    anchor = np_ref.enumerate_shifted_anchor(
        np_ref.generate_anchor_base(16, ratios, anchor_scales), 16, hh, ww)
    rpn_h, m_rpn = _relu(np_ref.conv2d_fwd(feat, P['rpn.conv1.W'], P['rpn.conv1.b'], 1, 1), 'rpn.conv1')
    rpn_out = np_ref.conv2d_fwd(rpn_h, P['rpn.loc_score.W'], P['rpn.loc_score.b'])
    nhwc = rpn_out.transpose(0, 2, 3, 1)
    rpn_locs = np.ascontiguousarray(nhwc[..., :4 * A]).reshape(N, -1, 4)
    rpn_scores = np.ascontiguousarray(nhwc[..., 4 * A:5 * A]).reshape(N, -1)
    pc = np_ref.ProposalCreator(**(proposal_creator_params or dict(min_size=0)))
    rois, roi_order = [], []
    for i in range(N):
        r, idx = pc(rpn_locs[i], rpn_scores[i], anchor, (H, W), float(scales[i]), train=True,
                    return_indices=True)
        rois.append(r)
        roi_order.append(idx)
    mark('rpn+proposals')
    # ---- targets (mask_rcnn_train_chain.py:126-158): every PTC call, then every ATC call ----
    ptc = np_targets.ProposalTargetCreator(n_sample=n_sample)
    s_rois, s_idx, g_locs, g_labels, g_masks = [], [], [], [], []
    for i in range(N):
        a, b, c, d = ptc(rois[i], bboxes[i], labels[i], masks[i])
        s_rois.append(a); s_idx.append(np.full(len(a), i, np.int32))
        g_locs.append(b); g_labels.append(c); g_masks.append(d)
    sample_rois = np.concatenate(s_rois, 0).astype(f32)
    sample_idx = np.concatenate(s_idx, 0)
    gt_roi_locs = np.concatenate(g_locs, 0).astype(f32)
    gt_roi_labels = np.concatenate(g_labels, 0)
    gt_roi_masks = np.concatenate(g_masks, 0)
    atc = np_ref.AnchorTargetCreator()
    r_locs, r_labels = zip(*[atc(b, anchor, (H, W)) for b in bboxes])
    gt_rpn_locs = np.concatenate(r_locs, 0).astype(f32)
    gt_rpn_labels = np.concatenate(r_labels, 0)
    mark('targets')
    # ---- RoI head (models/mask_rcnn_resnet.py:168-196) ----
    from . import roi_align_fwd, roi_align_bwd
    R = len(sample_rois)
    rois_xy = np.concatenate([sample_idx.astype(f32)[:, None], sample_rois], 1)[:, [0, 2, 1, 4, 3]]
    pool = roi_align_fwd(feat, rois_xy, roi_size, roi_size, 1. / 16, 0)
    mark('roi_align')
    res5, c_res5 = stage_fwd(pool, P, 'head.res5', 3, roi_size // 7)
    pool5 = np_ref.average_pooling_2d(res5, 7, 7)
    Wfc, bfc = P['head.cls_loc_score.W'][:5 * n_class], P['head.cls_loc_score.b'][:5 * n_class]
    fc = np_ref.linear_fwd(pool5, Wfc, bfc)
    roi_cls_locs, roi_scores = fc[:, :4 * n_class], fc[:, 4 * n_class:]
    dpre = np_ref.deconv2x2s2_fwd(res5, P['head.deconv6.W'], P['head.deconv6.b'])
    d6, m_d6 = _relu(dpre, 'head.deconv6')
    roi_masks = np_ref.conv2d_fwd(d6, P['head.mask.W'], P['head.mask.b'])
    mark('head')
    # ---- losses (:163-181) ----
    L = {}
    L['rpn_loc_loss'], g_rpn_loc = np_ref.fast_rcnn_loc_loss(
        rpn_locs.reshape(-1, 4), gt_rpn_locs, gt_rpn_labels, rpn_sigma)
    L['rpn_cls_loss'], g_rpn_score = np_ref.sigmoid_cross_entropy(rpn_scores.reshape(-1), gt_rpn_labels)
    ar = np.arange(R)
    sel_loc = roi_cls_locs.reshape(R, -1, 4)[ar, gt_roi_labels]
    L['roi_loc_loss'], g_sel_loc = np_ref.fast_rcnn_loc_loss(sel_loc, gt_roi_locs, gt_roi_labels, roi_sigma)
    L['roi_cls_loss'], g_scores = np_ref.softmax_cross_entropy(roi_scores, gt_roi_labels)
    sel_mask = roi_masks[ar, gt_roi_labels - 1]          # bg rows pick class -1, all ignored
    L['roi_mask_loss'], g_sel_mask = np_ref.sigmoid_cross_entropy(sel_mask, gt_roi_masks)
    L['loss'] = f32(sum(float(L[k]) for k in ('rpn_loc_loss', 'rpn_cls_loss', 'roi_loc_loss',
                                               'roi_cls_loss', 'roi_mask_loss')))
    mark('losses')
    out = dict(losses={k: float(v) for k, v in L.items()}, rois=rois, roi_order=roi_order,
               sample_rois=sample_rois, sample_roi_indices=sample_idx,
               gt_roi_labels=gt_roi_labels, gt_roi_masks=gt_roi_masks, gt_rpn_labels=gt_rpn_labels,
               feature_shape=feat.shape)
    if not backward:
        return out
    # ---- backward, reverse layer order ----
    G = {}
    g_masks_out = np.zeros_like(roi_masks)
    g_masks_out[ar, gt_roi_labels - 1] = g_sel_mask
    g_d6, G['head.mask.W'], G['head.mask.b'] = np_ref.conv2d_bwd(d6, P['head.mask.W'], g_masks_out)
    g_res5, G['head.deconv6.W'], G['head.deconv6.b'] = np_ref.deconv2x2s2_bwd(
        res5, P['head.deconv6.W'], g_d6 * m_d6)
    g_fc = np.zeros((R, P['head.cls_loc_score.W'].shape[0]), f32)
    g_cls = np.zeros((R, n_class, 4), f32)
    g_cls[ar, gt_roi_labels] = g_sel_loc
    g_fc[:, :4 * n_class] = g_cls.reshape(R, -1)
    g_fc[:, 4 * n_class:5 * n_class] = g_scores
    G['head.cls_loc_score.W'] = g_fc.T @ pool5.reshape(R, -1)
    G['head.cls_loc_score.b'] = g_fc.sum(0)
    g_pool5 = g_fc[:, :5 * n_class] @ Wfc
    g_res5 = g_res5 + np.broadcast_to((g_pool5 / f32(49.)).reshape(R, -1, 1, 1), res5.shape)
    g_pool = stage_bwd(g_res5, c_res5, P, 'head.res5', G)
    mark('head backward')
    g_feat = roi_align_bwd(g_pool, rois_xy, feat.shape, 1. / 16, 0)
    mark('roi_align backward')
    g_out = np.zeros_like(rpn_out)
    g_nhwc = g_out.transpose(0, 2, 3, 1)
    g_nhwc[..., :4 * A] = g_rpn_loc.reshape(N, hh, ww, 4 * A)
    g_nhwc[..., 4 * A:5 * A] = g_rpn_score.reshape(N, hh, ww, A)
    g_rpn_h, G['rpn.loc_score.W'], G['rpn.loc_score.b'] = np_ref.conv2d_bwd(
        rpn_h, P['rpn.loc_score.W'], g_out)
    g_f2, G['rpn.conv1.W'], G['rpn.conv1.b'] = np_ref.conv2d_bwd(
        feat, P['rpn.conv1.W'], g_rpn_h * m_rpn, 1, 1)
    g_feat = g_feat + g_f2
    mark('rpn backward')
    g_res3 = stage_bwd(g_feat, c_res4, P, 'extractor.res4', G)
    stage_bwd(g_res3, c_res3, P, 'extractor.res3', G, need_gx=False)   # unchain_backward at res2
    mark('extractor backward')
    out['grads'] = G
    return out


# ---------------------------------------------------------------------------------------
# deterministic synthetic parameters / inputs (no trained weights offline)
# ---------------------------------------------------------------------------------------
def param_shapes(n_layers=50, n_class=81, n_anchor=15):
    """{name: logical shape} of MaskRCNNResNet (models/mask_rcnn_resnet.py:30-143)."""
    nb = BLOCKS[n_layers]
    S = {'extractor.conv1.W': (64, 3, 7, 7), 'extractor.conv1.b': (64,),
         'extractor.bn1.W': (64,), 'extractor.bn1.b': (64,)}

    def stage(pre, n, cin, mid, cout):
        for i in range(n):
            b = pre + ('.a' if i == 0 else '.b%d' % i)
            ci = cin if i == 0 else cout
            S[b + '.conv1.W'] = (mid, ci, 1, 1)
            S[b + '.conv2.W'] = (mid, mid, 3, 3)
            S[b + '.conv3.W'] = (cout, mid, 1, 1)
            for k, c in (('bn1', mid), ('bn2', mid), ('bn3', cout)):
                S[b + '.%s.W' % k] = (c,)
                S[b + '.%s.b' % k] = (c,)
            if i == 0:
                S[b + '.conv4.W'] = (cout, ci, 1, 1)
                S[b + '.bn4.W'] = (cout,)
                S[b + '.bn4.b'] = (cout,)
    stage('extractor.res2', nb[0], 64, 64, 256)
    stage('extractor.res3', nb[1], 256, 128, 512)
    stage('extractor.res4', nb[2], 512, 256, 1024)
    S['rpn.conv1.W'] = (1024, 1024, 3, 3); S['rpn.conv1.b'] = (1024,)
    n_rpn = (5 * n_anchor + 3) // 4 * 4
    S['rpn.loc_score.W'] = (n_rpn, 1024, 1, 1); S['rpn.loc_score.b'] = (n_rpn,)
    stage('head.res5', 3, 1024, 512, 2048)
    n_fc = (5 * n_class + 3) // 4 * 4
    S['head.cls_loc_score.W'] = (n_fc, 2048); S['head.cls_loc_score.b'] = (n_fc,)
    S['head.deconv6.W'] = (2048, 256, 2, 2); S['head.deconv6.b'] = (256,)
    S['head.mask.W'] = (n_class - 1, 256, 1, 1); S['head.mask.b'] = (n_class - 1,)
    return S


def synthetic_params(n_layers=50, n_class=81, n_anchor=15, seed=0):
    """Seeded parameters with O(1) activations through the residual chains: He-normal filters,
    affine scales < 1 on the residual branches, small heads (the reference's initialisers,
    models/mask_rcnn_resnet.py:57-64).  One RandomState per parameter name, so any subset can
    be regenerated independently (the GPU test fills its model from the same recipe)."""
    import zlib
    P = {}
    for name, shape in param_shapes(n_layers, n_class, n_anchor).items():
        rng = np.random.RandomState((zlib.crc32(name.encode()) + seed) % (2 ** 31))
        leaf = name.rsplit('.', 2)[-2]
        if name.endswith('.b'):
            v = rng.standard_normal(shape) * 0.05
        elif leaf.startswith('bn'):
            lo, hi = {'bn3': (0.2, 0.35), 'bn4': (0.4, 0.6)}.get(leaf, (0.6, 1.0))
            if name == 'extractor.bn1.W':
                lo, hi = 1. / 80, 1. / 50          # mean-subtracted images are O(128)
            v = rng.uniform(lo, hi, shape)
        elif name in ('rpn.loc_score.W', 'head.mask.W', 'head.deconv6.W'):
            v = rng.standard_normal(shape) * 0.01
        elif name == 'head.cls_loc_score.W':
            v = rng.standard_normal(shape) * 0.01
            v[:4 * n_class] *= 0.1
            v[5 * n_class:] = 0.
        else:
            fan_in = int(np.prod(shape[1:]))
            v = rng.standard_normal(shape) * np.sqrt(2. / fan_in)
        P[name] = v.astype(f32)
    P['rpn.loc_score.W'][5 * n_anchor:] = 0.
    P['rpn.loc_score.b'][5 * n_anchor:] = 0.
    P['head.cls_loc_score.b'][5 * n_class:] = 0.
    return P


def synthetic_inputs(seed, batch, H, W, n_gt=8, n_fg_class=80, scale=1.6):
    """bench.py's synthetic batch."""
    rng = np.random.RandomState(seed)
    mean = np.asarray((123.152, 115.903, 103.063), f32)[:, None, None]
    imgs = (rng.uniform(0, 255, (batch, 3, H, W)).astype(f32) - mean)
    bboxes, labels, masks = [], [], []
    yy, xx = np.mgrid[0:H, 0:W]
    for _ in range(batch):
        hh = rng.uniform(32, min(400, H * 0.6), n_gt)
        ww = rng.uniform(32, min(400, W * 0.6), n_gt)
        y0 = rng.uniform(0, H - 32, n_gt)
        x0 = rng.uniform(0, W - 32, n_gt)
        b = np.stack([y0, x0, np.minimum(y0 + hh, H), np.minimum(x0 + ww, W)], 1).astype(f32)
        m = np.zeros((n_gt, H, W), np.int32)
        for g in range(n_gt):
            cy, cx = (b[g, 0] + b[g, 2]) / 2, (b[g, 1] + b[g, 3]) / 2
            ry, rx = (b[g, 2] - b[g, 0]) / 2, (b[g, 3] - b[g, 1]) / 2
            m[g] = (((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0)
        bboxes.append(b)
        labels.append(rng.randint(0, n_fg_class, n_gt).astype(np.int32))
        masks.append(m)
    return imgs, bboxes, labels, masks, np.full((batch,), scale, f32)
