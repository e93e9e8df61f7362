"""Plain PyTorch CPU reference of the model graph (torch.nn.functional ops + autograd),
used only as a checker for the HIP path's end-to-end forward/backward.  It runs in the dtype
of ``RefParams`` — float64 by default, so that a comparison bounds the HIP path's OWN fp32
error rather than the difference of two fp32 roundings.  ROIAlign goes through the oracle
(fp32 C restatement, fwd/bwd) wrapped as an autograd Function."""
import numpy as np
import torch
import torch.nn.functional as F

import oracle


class _RefROIAlign(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, rois_xy, outh, outw, scale):
        y = oracle.roi_align_fwd(x.detach().numpy().astype(np.float32),
                                 rois_xy.numpy().astype(np.float32), outh, outw, scale, 0)
        ctx.save_for_backward(rois_xy)
        ctx.meta = (tuple(x.shape), scale, x.dtype)
        return torch.tensor(y, dtype=x.dtype)

    @staticmethod
    def backward(ctx, gy):
        rois_xy, = ctx.saved_tensors
        shape, scale, dtype = ctx.meta
        gx = oracle.roi_align_bwd(gy.contiguous().numpy().astype(np.float32),
                                  rois_xy.numpy().astype(np.float32), shape, scale, 0)
        return torch.tensor(gx, dtype=dtype), None, None, None, None


class RefParams(object):
    """CPU leaf copies (logical NCHW-contiguous) of a HIP model's parameters."""

    def __init__(self, module, dtype=torch.float64):
        self.p = {}
        self.dtype = dtype
        for name, t in module.named_parameters():
            c = t.detach().cpu().contiguous().clone().to(dtype)
            c.requires_grad_(True)
            self.p[name] = c

    def __getitem__(self, k):
        return self.p[k]


def _affine(x, P, pre):
    return x * P[pre + '.W'].detach().view(1, -1, 1, 1) + P[pre + '.b'].detach().view(1, -1, 1, 1)


class Relus(object):
    """ReLU bookkeeping of the reference graph.  ``masks`` {name: bool tensor}: use these
    decisions instead of the graph's own (y = x * mask); ``pre``: filled with every ReLU's float64
    pre-activation (detached) when it is a dict.  Names: '<bottleneck>.h1' / '.h2' / '.y',
    'rpn.conv1', 'head.deconv6'."""

    def __init__(self, masks=None, record=False):
        self.masks = masks
        self.pre = {} if record else None

    def __call__(self, x, name):
        if self.pre is not None:
            self.pre[name] = x.detach()
        if self.masks is not None and name in self.masks:
            return x * self.masks[name].to(x.dtype)
        return F.relu(x)


_PLAIN = Relus()


def bottleneck(x, P, pre, stride, proj, relus=_PLAIN):
    h = relus(_affine(F.conv2d(x, P[pre + '.conv1.W'], stride=stride), P, pre + '.bn1'), pre + '.h1')
    h = relus(_affine(F.conv2d(h, P[pre + '.conv2.W'], padding=1), P, pre + '.bn2'), pre + '.h2')
    h = _affine(F.conv2d(h, P[pre + '.conv3.W']), P, pre + '.bn3')
    sc = _affine(F.conv2d(x, P[pre + '.conv4.W'], stride=stride), P, pre + '.bn4') if proj else x
    return relus(h + sc, pre + '.y')


def building_block(x, P, pre, n, stride, relus=_PLAIN):
    x = bottleneck(x, P, pre + '.a', stride, True, relus)
    for i in range(1, n):
        x = bottleneck(x, P, pre + '.b%d' % i, 1, False, relus)
    return x


def extractor(x, P, pre='extractor', blocks=(3, 4, 6), relus=_PLAIN):
    x = x.to(P.dtype)
    with torch.no_grad():
        h = F.conv2d(x, P[pre + '.conv1.W'], P[pre + '.conv1.b'], stride=2, padding=3)
        h = F.relu(_affine(h, P, pre + '.bn1'))
        h = F.max_pool2d(h, 3, 2, 1, ceil_mode=True)
        h = building_block(h, P, pre + '.res2', blocks[0], 1)
    h = h.detach()
    h = building_block(h, P, pre + '.res3', blocks[1], 2, relus)
    h = building_block(h, P, pre + '.res4', blocks[2], 2, relus)
    return h


def rpn(feat, P, A, pre='rpn', relus=_PLAIN):
    h = relus(F.conv2d(feat, P[pre + '.conv1.W'], P[pre + '.conv1.b'], padding=1), pre + '.conv1')
    out = F.conv2d(h, P[pre + '.loc_score.W'], P[pre + '.loc_score.b'])
    n = feat.shape[0]
    nhwc = out.permute(0, 2, 3, 1)
    locs = nhwc[..., :4 * A].reshape(n, -1, 4)
    scores = nhwc[..., 4 * A:5 * A].reshape(n, -1)
    return locs, scores


def head(feat, rois_yx, roi_indices, P, n_class, roi_size, pre='head', relus=_PLAIN):
    rois = torch.cat([roi_indices.float()[:, None], rois_yx.float()], 1)[:, [0, 2, 1, 4, 3]].contiguous()
    pool = _RefROIAlign.apply(feat, rois, roi_size, roi_size, 1. / 16)
    res5 = building_block(pool, P, pre + '.res5', 3, roi_size // 7, relus)
    pool5 = F.avg_pool2d(res5, 7, 7).flatten(1)
    fc = F.linear(pool5, P[pre + '.cls_loc_score.W'], P[pre + '.cls_loc_score.b'])
    cls_locs, scores = fc[:, :4 * n_class], fc[:, 4 * n_class:5 * n_class]
    d = relus(F.conv_transpose2d(res5, P[pre + '.deconv6.W'], P[pre + '.deconv6.b'], stride=2),
              pre + '.deconv6')
    masks = F.conv2d(d, P[pre + '.mask.W'], P[pre + '.mask.b'])
    return cls_locs, scores, masks


def losses(rpn_locs, rpn_scores, gt_rpn_locs, gt_rpn_labels, cls_locs, scores, masks,
           gt_roi_locs, gt_roi_labels, gt_roi_masks, rpn_sigma=3., roi_sigma=1.):
    """models/mask_rcnn_train_chain.py:163-181 in plain torch."""
    def loc_loss(pred, gt, label, sigma):
        s2 = sigma ** 2
        w = (label > 0).to(pred.dtype)[:, None]
        d = w * (pred - gt.to(pred.dtype))
        a = d.abs()
        flag = (a.detach() < 1. / s2).to(pred.dtype)
        y = flag * (s2 / 2.) * d * d + (1 - flag) * (a - 0.5 / s2)
        return y.sum() / (label >= 0).sum().to(pred.dtype)

    def sce(x, t):
        m = t != -1
        cnt = max(int(m.sum()), 1)
        return F.binary_cross_entropy_with_logits(x[m], t[m].to(x.dtype), reduction='sum') / cnt

    n = len(cls_locs)
    rl = loc_loss(rpn_locs.reshape(-1, 4), gt_rpn_locs, gt_rpn_labels, rpn_sigma)
    rc = sce(rpn_scores.reshape(-1), gt_rpn_labels)
    sel = cls_locs.reshape(n, -1, 4)[torch.arange(n), gt_roi_labels.long()]
    ol = loc_loss(sel, gt_roi_locs, gt_roi_labels, roi_sigma)
    oc = F.cross_entropy(scores, gt_roi_labels.long(), ignore_index=-1)
    msel = masks[torch.arange(n), (gt_roi_labels.long() - 1)]
    om = sce(msel, gt_roi_masks)
    return rl, rc, ol, oc, om
