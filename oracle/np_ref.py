"""ORACLE (test infrastructure, not product code) — NumPy restatements.

Restates, in plain NumPy fp32, the third-party (chainer / chainercv) pieces of
the reference hot path whose source is NOT under /root/reference (un-vendored
dependencies: requirements.txt:1-2 `chainer>=4.0.0,!=5.0.0,!=5.1.0`,
`chainercv>=0.9.0`, no lock file) and the reference's own small pieces.  Each
function cites the reference call site it serves and the algorithm statement
it follows.

"parity unpinned": the reference holds no golden vectors for any of these
(SURVEY.md section 4); they are pinned by independent known-answer tests in
tests/test_oracle_*.py (torch.nn.functional on CPU for conv/pool/linear/loss,
brute-force definitions for NMS / IoU).

Box convention everywhere: (y_min, x_min, y_max, x_max), fp32.
"""
import numpy as np

f32 = np.float32


# --------------------------------------------------------------------------
# chainercv box utilities (Appendix A.2)
# --------------------------------------------------------------------------

def generate_anchor_base(base_size=16, ratios=(0.5, 1, 2),
                         anchor_scales=(8, 16, 32)):
    """call site: models/region_proposal_network.py:67-68."""
    py = base_size / 2.
    px = base_size / 2.
    anchor_base = np.zeros((len(ratios) * len(anchor_scales), 4), dtype=f32)
    for i in range(len(ratios)):
        for j in range(len(anchor_scales)):
            h = base_size * anchor_scales[j] * np.sqrt(ratios[i])
            w = base_size * anchor_scales[j] * np.sqrt(1. / ratios[i])
            index = i * len(anchor_scales) + j
            anchor_base[index, 0] = py - h / 2.
            anchor_base[index, 1] = px - w / 2.
            anchor_base[index, 2] = py + h / 2.
            anchor_base[index, 3] = px + w / 2.
    return anchor_base


def enumerate_shifted_anchor(anchor_base, feat_stride, height, width):
    """models/region_proposal_network.py:148-167 (reference code)."""
    shift_y = np.arange(0, height * feat_stride, feat_stride)
    shift_x = np.arange(0, width * feat_stride, feat_stride)
    shift_x, shift_y = np.meshgrid(shift_x, shift_y)
    shift = np.stack((shift_y.ravel(), shift_x.ravel(),
                      shift_y.ravel(), shift_x.ravel()), axis=1)
    A = anchor_base.shape[0]
    K = shift.shape[0]
    anchor = anchor_base.reshape((1, A, 4)) + \
        shift.reshape((1, K, 4)).transpose((1, 0, 2))
    return anchor.reshape((K * A, 4)).astype(f32)


def loc2bbox(src_bbox, loc):
    """call sites: models/mask_rcnn.py:232, ProposalCreator (A.4).

    exp() is evaluated in float64 and rounded to fp32 (correctly rounded up to
    double-rounding); the HIP decode kernel does the same so that decoded boxes
    — and therefore NMS keep sets — are bit-identical.  NumPy's own float32 exp
    is SIMD-implementation dependent (<1 ulp) so no stronger statement about
    the literal reference is possible.
    """
    if src_bbox.shape[0] == 0:
        return np.zeros((0, 4), dtype=loc.dtype)
    src_bbox = src_bbox.astype(f32, copy=False)
    loc = loc.astype(f32, copy=False)
    src_h = src_bbox[:, 2] - src_bbox[:, 0]
    src_w = src_bbox[:, 3] - src_bbox[:, 1]
    src_cy = src_bbox[:, 0] + f32(0.5) * src_h
    src_cx = src_bbox[:, 1] + f32(0.5) * src_w
    dy, dx, dh, dw = loc[:, 0], loc[:, 1], loc[:, 2], loc[:, 3]
    cy = dy * src_h + src_cy
    cx = dx * src_w + src_cx
    h = np.exp(dh.astype(np.float64)).astype(f32) * src_h
    w = np.exp(dw.astype(np.float64)).astype(f32) * src_w
    dst = np.zeros(loc.shape, dtype=f32)
    dst[:, 0] = cy - f32(0.5) * h
    dst[:, 1] = cx - f32(0.5) * w
    dst[:, 2] = cy + f32(0.5) * h
    dst[:, 3] = cx + f32(0.5) * w
    return dst


def bbox2loc(src_bbox, dst_bbox):
    """call sites: models/utils/proposal_target_creator.py:156, AnchorTargetCreator."""
    src_bbox = src_bbox.astype(f32, copy=False)
    dst_bbox = dst_bbox.astype(f32, copy=False)
    height = src_bbox[:, 2] - src_bbox[:, 0]
    width = src_bbox[:, 3] - src_bbox[:, 1]
    ctr_y = src_bbox[:, 0] + f32(0.5) * height
    ctr_x = src_bbox[:, 1] + f32(0.5) * width
    base_height = dst_bbox[:, 2] - dst_bbox[:, 0]
    base_width = dst_bbox[:, 3] - dst_bbox[:, 1]
    base_ctr_y = dst_bbox[:, 0] + f32(0.5) * base_height
    base_ctr_x = dst_bbox[:, 1] + f32(0.5) * base_width
    eps = np.finfo(f32).eps
    height = np.maximum(height, eps)
    width = np.maximum(width, eps)
    dy = (base_ctr_y - ctr_y) / height
    dx = (base_ctr_x - ctr_x) / width
    dh = np.log(base_height / height)
    dw = np.log(base_width / width)
    return np.vstack((dy, dx, dh, dw)).transpose().astype(f32)


def bbox_iou(bbox_a, bbox_b):
    """call site: models/utils/proposal_target_creator.py:124."""
    if bbox_a.shape[1] != 4 or bbox_b.shape[1] != 4:
        raise IndexError
    tl = np.maximum(bbox_a[:, None, :2], bbox_b[:, :2])
    br = np.minimum(bbox_a[:, None, 2:], bbox_b[:, 2:])
    area_i = np.prod(br - tl, axis=2) * (tl < br).all(axis=2)
    area_a = np.prod(bbox_a[:, 2:] - bbox_a[:, :2], axis=1)
    area_b = np.prod(bbox_b[:, 2:] - bbox_b[:, :2], axis=1)
    return area_i / (area_a[:, None] + area_b - area_i)


def stable_argsort_desc(score):
    """Descending score, ascending index on ties — the build's documented tie
    rule (NumPy's default argsort is unstable, so the reference's tie order is
    unspecified; Appendix A.3)."""
    return np.argsort(-score.astype(f32), kind='stable')


def non_maximum_suppression(bbox, thresh, score=None, limit=None):
    """chainercv CPU NMS (Appendix A.3); call sites models/mask_rcnn.py:193-194
    and ProposalCreator."""
    if len(bbox) == 0:
        return np.zeros((0,), dtype=np.int32)
    bbox = bbox.astype(f32, copy=False)
    if score is not None:
        order = stable_argsort_desc(score)
        bbox = bbox[order]
    bbox_area = np.prod(bbox[:, 2:] - bbox[:, :2], axis=1)
    selec = np.zeros(bbox.shape[0], dtype=bool)
    for i, b in enumerate(bbox):
        tl = np.maximum(b[:2], bbox[selec, :2])
        br = np.minimum(b[2:], bbox[selec, 2:])
        area = np.prod(br - tl, axis=1) * (tl < br).all(axis=1)
        iou = area / (bbox_area[i] + bbox_area[selec] - area)
        if (iou >= thresh).any():
            continue
        selec[i] = True
        if limit is not None and np.count_nonzero(selec) >= limit:
            break
    selec = np.where(selec)[0]
    if score is not None:
        selec = order[selec]
    return selec.astype(np.int32)


class ProposalCreator(object):
    """chainercv ProposalCreator (Appendix A.4); ctor call
    models/region_proposal_network.py:70 with params
    models/mask_rcnn_resnet.py:48-52; per-image call :135-138."""

    def __init__(self, nms_thresh=0.7, n_train_pre_nms=12000,
                 n_train_post_nms=2000, n_test_pre_nms=6000,
                 n_test_post_nms=300, force_cpu_nms=False, min_size=16):
        self.nms_thresh = nms_thresh
        self.n_train_pre_nms = n_train_pre_nms
        self.n_train_post_nms = n_train_post_nms
        self.n_test_pre_nms = n_test_pre_nms
        self.n_test_post_nms = n_test_post_nms
        self.force_cpu_nms = force_cpu_nms
        self.min_size = min_size

    def __call__(self, loc, score, anchor, img_size, scale=1., train=True,
                 return_indices=False):
        if train:
            n_pre_nms, n_post_nms = self.n_train_pre_nms, self.n_train_post_nms
        else:
            n_pre_nms, n_post_nms = self.n_test_pre_nms, self.n_test_post_nms
        roi = loc2bbox(anchor, loc)
        roi[:, slice(0, 4, 2)] = np.clip(roi[:, slice(0, 4, 2)], 0, img_size[0])
        roi[:, slice(1, 4, 2)] = np.clip(roi[:, slice(1, 4, 2)], 0, img_size[1])
        min_size = f32(self.min_size * scale)
        hs = roi[:, 2] - roi[:, 0]
        ws = roi[:, 3] - roi[:, 1]
        keep = np.where((hs >= min_size) & (ws >= min_size))[0]
        roi = roi[keep, :]
        score = score[keep]
        order = stable_argsort_desc(score.ravel())
        if n_pre_nms > 0:
            order = order[:n_pre_nms]
        roi = roi[order, :]
        keep2 = non_maximum_suppression(roi, thresh=self.nms_thresh)
        if n_post_nms > 0:
            keep2 = keep2[:n_post_nms]
        out = roi[keep2]
        if return_indices:
            return out, keep[order][keep2].astype(np.int32)
        return out


class AnchorTargetCreator(object):
    """chainercv AnchorTargetCreator (Appendix A.5); default instance
    models/mask_rcnn_train_chain.py:61, call :153-158.  Consumes the global
    np.random stream (seeded at examples/train_common.py:135-136)."""

    def __init__(self, n_sample=256, pos_iou_thresh=0.7, neg_iou_thresh=0.3,
                 pos_ratio=0.5):
        self.n_sample = n_sample
        self.pos_iou_thresh = pos_iou_thresh
        self.neg_iou_thresh = neg_iou_thresh
        self.pos_ratio = pos_ratio

    def __call__(self, bbox, anchor, img_size):
        img_H, img_W = img_size
        n_anchor = len(anchor)
        inside_index = np.where(
            (anchor[:, 0] >= 0) & (anchor[:, 1] >= 0) &
            (anchor[:, 2] <= img_H) & (anchor[:, 3] <= img_W))[0]
        anchor = anchor[inside_index]
        argmax_ious, label = self._create_label(inside_index, anchor, bbox)
        loc = bbox2loc(anchor, bbox[argmax_ious])
        label = _unmap(label, n_anchor, inside_index, fill=-1)
        loc = _unmap(loc, n_anchor, inside_index, fill=0)
        return loc, label

    def _create_label(self, inside_index, anchor, bbox):
        label = np.empty((len(inside_index),), dtype=np.int32)
        label.fill(-1)
        ious = bbox_iou(anchor, bbox)
        argmax_ious = ious.argmax(axis=1)
        max_ious = ious[np.arange(len(inside_index)), argmax_ious]
        gt_argmax_ious = ious.argmax(axis=0)
        gt_max_ious = ious[gt_argmax_ious, np.arange(ious.shape[1])]
        gt_argmax_ious = np.where(ious == gt_max_ious)[0]
        label[max_ious < self.neg_iou_thresh] = 0
        label[gt_argmax_ious] = 1
        label[max_ious >= self.pos_iou_thresh] = 1
        n_pos = int(self.pos_ratio * self.n_sample)
        pos_index = np.where(label == 1)[0]
        if len(pos_index) > n_pos:
            disable_index = np.random.choice(
                pos_index, size=(len(pos_index) - n_pos), replace=False)
            label[disable_index] = -1
        n_neg = self.n_sample - np.sum(label == 1)
        neg_index = np.where(label == 0)[0]
        if len(neg_index) > n_neg:
            disable_index = np.random.choice(
                neg_index, size=(len(neg_index) - n_neg), replace=False)
            label[disable_index] = -1
        return argmax_ious, label


def _unmap(data, count, index, fill=0):
    if len(data.shape) == 1:
        ret = np.empty((count,), dtype=data.dtype)
        ret.fill(fill)
        ret[index] = data
    else:
        ret = np.empty((count,) + data.shape[1:], dtype=data.dtype)
        ret.fill(fill)
        ret[index, :] = data
    return ret


# --------------------------------------------------------------------------
# chainer layers (Appendix A.1) — NCHW, fp32
# --------------------------------------------------------------------------

def conv_outsize(size, k, s, p, cover_all=False):
    if cover_all:
        return (size + p * 2 - k + s - 1) // s + 1
    return (size + p * 2 - k) // s + 1


def im2col(x, kh, kw, sy, sx, ph, pw, cover_all=False, pval=0.):
    n, c, h, w = x.shape
    out_h = conv_outsize(h, kh, sy, ph, cover_all)
    out_w = conv_outsize(w, kw, sx, pw, cover_all)
    img = np.pad(x, ((0, 0), (0, 0), (ph, ph + sy - 1), (pw, pw + sx - 1)),
                 mode='constant', constant_values=(pval,))
    col = np.ndarray((n, c, kh, kw, out_h, out_w), dtype=x.dtype)
    for j in range(kh):
        jl = j + sy * out_h
        for i in range(kw):
            il = i + sx * out_w
            col[:, :, j, i, :, :] = img[:, :, j:jl:sy, i:il:sx]
    return col


def col2im(col, sy, sx, ph, pw, h, w):
    n, c, kh, kw, out_h, out_w = col.shape
    img = np.zeros((n, c, h + 2 * ph + sy - 1, w + 2 * pw + sx - 1),
                   dtype=col.dtype)
    for j in range(kh):
        jl = j + sy * out_h
        for i in range(kw):
            il = i + sx * out_w
            img[:, :, j:jl:sy, i:il:sx] += col[:, :, j, i]
    return img[:, :, ph:h + ph, pw:w + pw]


def conv2d_fwd(x, W, b=None, stride=1, pad=0):
    """L.Convolution2D forward: cross-correlation, W (out,in,kh,kw)."""
    kh, kw = W.shape[2:]
    col = im2col(x, kh, kw, stride, stride, pad, pad)
    y = np.tensordot(col, W, ((1, 2, 3), (1, 2, 3))).astype(x.dtype, copy=False)
    if b is not None:
        y += b
    return np.rollaxis(y, 3, 1)


def conv2d_bwd(x, W, gy, stride=1, pad=0, need_gx=True):
    """returns gx, gW, gb."""
    kh, kw = W.shape[2:]
    h, w = x.shape[2:]
    col = im2col(x, kh, kw, stride, stride, pad, pad)
    gW = np.tensordot(gy, col, ((0, 2, 3), (0, 4, 5))).astype(W.dtype, copy=False)
    gb = gy.sum(axis=(0, 2, 3))
    gx = None
    if need_gx:
        gcol = np.tensordot(W, gy, (0, 1)).astype(x.dtype, copy=False)
        gcol = np.rollaxis(gcol, 3)
        gx = col2im(gcol, stride, stride, pad, pad, h, w)
    return gx, gW, gb


def deconv2x2s2_fwd(x, W, b=None):
    """L.Deconvolution2D(in,out,ksize=2,stride=2): W (in,out,2,2).
    y[n,o,2i+a,2j+b] = sum_c x[n,c,i,j] W[c,o,a,b] + bias[o]."""
    n, c, h, w = x.shape
    o = W.shape[1]
    y = np.einsum('ncij,coab->noiajb', x, W, optimize=True)
    y = y.reshape(n, o, 2 * h, 2 * w).astype(f32)
    if b is not None:
        y += b[None, :, None, None]
    return y


def deconv2x2s2_bwd(x, W, gy):
    n, c, h, w = x.shape
    o = W.shape[1]
    g = gy.reshape(n, o, h, 2, w, 2)
    gx = np.einsum('noiajb,coab->ncij', g, W, optimize=True).astype(f32)
    gW = np.einsum('ncij,noiajb->coab', x, g, optimize=True).astype(f32)
    gb = gy.sum(axis=(0, 2, 3))
    return gx, gW, gb


def max_pooling_2d(x, k=3, stride=2, pad=1, cover_all=True):
    """F.max_pooling_2d(x,3,stride=2,pad=1) — cover_all=True default, pad -inf
    (models/resnet_extractor.py:69)."""
    col = im2col(x, k, k, stride, stride, pad, pad, cover_all=cover_all,
                 pval=-np.inf)
    n, c, kh, kw, oh, ow = col.shape
    return col.reshape(n, c, kh * kw, oh, ow).max(axis=2)


def average_pooling_2d(x, k, stride):
    """F.average_pooling_2d(res5, 7, stride=7) (models/mask_rcnn_resnet.py:188)."""
    col = im2col(x, k, k, stride, stride, 0, 0)
    return col.mean(axis=(2, 3)).astype(f32)


def linear_fwd(x, W, b):
    return (x.reshape(len(x), -1) @ W.T + b).astype(x.dtype, copy=False)


def affine_channel_2d_fwd(x, W, b):
    """functions/affine_channel_2d.py:10-22: y = W*x + b, W,b broadcast (1,C,1,1)."""
    return W.reshape(1, -1, 1, 1) * x + b.reshape(1, -1, 1, 1)


def affine_channel_2d_bwd(x, W, gy):
    """functions/affine_channel_2d.py:38-56."""
    gx = W.reshape(1, -1, 1, 1) * gy
    gW = (x * gy).sum(axis=(0, 2, 3))
    gb = gy.sum(axis=(0, 2, 3))
    return gx, gW, gb


# --------------------------------------------------------------------------
# losses (Appendix A.1; models/mask_rcnn_train_chain.py:163-181,192-213)
# --------------------------------------------------------------------------

def sigmoid_cross_entropy(x, t):
    """F.sigmoid_cross_entropy(x, t, normalize=True); t in {-1,0,1}, -1 ignored.
    returns (loss, gx)."""
    x = x.astype(f32)
    ignore = (t == -1)
    count = max(int((~ignore).sum()), 1)
    valid = (~ignore).astype(f32)
    loss_el = -valid * (x * (t - (x >= 0)) - np.log1p(np.exp(-np.abs(x))))
    loss = f32(loss_el.sum(dtype=np.float64) / count)
    sig = 1. / (1. + np.exp(-x.astype(np.float64)))
    gx = (valid * (sig - t) / count).astype(f32)
    return loss, gx


def softmax_cross_entropy(x, t):
    """F.softmax_cross_entropy(x, t), ignore_label=-1, mean over valid rows."""
    x64 = x.astype(np.float64)
    m = x64.max(axis=1, keepdims=True)
    logz = m + np.log(np.exp(x64 - m).sum(axis=1, keepdims=True))
    logp = x64 - logz
    valid = (t != -1)
    count = max(int(valid.sum()), 1)
    idx = np.where(valid, t, 0)
    loss = f32(-(logp[np.arange(len(t)), idx] * valid).sum() / count)
    g = np.exp(logp)
    g[np.arange(len(t)), idx] -= 1
    g *= valid[:, None] / count
    return loss, g.astype(f32)


def fast_rcnn_loc_loss(pred_loc, gt_loc, gt_label, sigma):
    """_fast_rcnn_loc_loss + _smooth_l1_loss, models/mask_rcnn_train_chain.py:192-213."""
    sigma2 = sigma ** 2
    in_weight = np.zeros_like(gt_loc)
    in_weight[gt_label > 0] = 1
    diff = in_weight * (pred_loc - gt_loc)
    abs_diff = np.abs(diff)
    flag = (abs_diff < (1. / sigma2)).astype(f32)
    y = flag * (sigma2 / 2.) * np.square(diff) + \
        (1 - flag) * (abs_diff - 0.5 / sigma2)
    n = (gt_label >= 0).sum()
    loss = f32(y.sum(dtype=np.float64) / n)
    g = (in_weight * (flag * sigma2 * diff + (1 - flag) * np.sign(diff)) / n)
    return loss, g.astype(f32)


def momentum_sgd_wd(p, g, v, lr, momentum=0.9, wd=1e-4):
    """chainer WeightDecay hook + MomentumSGD rule (Appendix A.1;
    examples/train_common.py:176-180): g += wd*p; v = m*v - lr*g; p += v."""
    g = g + f32(wd) * p
    v = f32(momentum) * v - f32(lr) * g
    return p + v, v


# --------------------------------------------------------------------------
# cv2.resize(INTER_LINEAR) restatement used by ProposalTargetCreator
# (models/utils/proposal_target_creator.py:171-172).  cv2 is not installed
# here: restated from OpenCV's documented half-pixel-centre bilinear rule
# (src = (dst + 0.5) * scale - 0.5, clamped to the border).
# --------------------------------------------------------------------------

def resize_bilinear(img, out_h, out_w):
    """img (h, w) or (h, w, c) float32 -> (out_h, out_w[, c]) float32."""
    img = np.asarray(img, f32)
    h, w = img.shape[:2]

    def coords(n_out, n_in):
        scale = n_in / float(n_out)
        s = (np.arange(n_out, dtype=np.float64) + 0.5) * scale - 0.5
        i0 = np.floor(s).astype(np.int64)
        frac = (s - i0).astype(f32)
        lo = i0 < 0
        frac[lo] = 0.
        i0[lo] = 0
        hi = i0 >= n_in - 1
        frac[hi] = 0.
        i0[hi] = n_in - 1
        i1 = np.minimum(i0 + 1, n_in - 1)
        return i0, i1, frac

    y0, y1, fy = coords(out_h, h)
    x0, x1, fx = coords(out_w, w)
    if img.ndim == 3:
        fy_ = fy[:, None, None]
        fx_ = fx[None, :, None]
    else:
        fy_ = fy[:, None]
        fx_ = fx[None, :]
    top = img[y0][:, x0] * (1 - fx_) + img[y0][:, x1] * fx_
    bot = img[y1][:, x0] * (1 - fx_) + img[y1][:, x1] * fx_
    return (top * (1 - fy_) + bot * fy_).astype(f32)
