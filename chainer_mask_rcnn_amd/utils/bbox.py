"""Host-side (NumPy) box utilities used by the target creators.

The reference takes these from chainercv (``generate_anchor_base``, ``bbox_iou``,
``bbox2loc``; imports at /root/reference/chainer_mask_rcnn/models/region_proposal_network.py:20-23
and models/utils/proposal_target_creator.py:19-20) and runs them on the CPU with
NumPy as well (proposal_target_creator.py:112-115), so they stay host code here.
Boxes are (y_min, x_min, y_max, x_max) float32.
"""
import numpy as np


def generate_anchor_base(base_size=16, ratios=(0.5, 1, 2), anchor_scales=(8, 16, 32)):
    """(len(ratios)*len(scales), 4) anchors centred on (base/2, base/2), ratio-major."""
    r = np.asarray(ratios, dtype=np.float64)[:, None]
    s = np.asarray(anchor_scales, dtype=np.float64)[None, :]
    h = (base_size * s * np.sqrt(r)).ravel()
    w = (base_size * s * np.sqrt(1. / r)).ravel()
    c = base_size / 2.
    return np.stack([c - h / 2., c - w / 2., c + h / 2., c + w / 2.], axis=1).astype(np.float32)


def enumerate_shifted_anchor(anchor_base, feat_stride, height, width):
    """All anchors of an (height, width) feature map, cell-major then anchor
    (models/region_proposal_network.py:148-167)."""
    ys = np.arange(0, height * feat_stride, feat_stride)
    xs = np.arange(0, width * feat_stride, feat_stride)
    gy, gx = np.meshgrid(ys, xs, indexing='ij')
    shift = np.stack([gy.ravel(), gx.ravel(), gy.ravel(), gx.ravel()], axis=1)
    anchor = shift[:, None, :] + anchor_base[None, :, :]
    return anchor.reshape(-1, 4).astype(np.float32)


def bbox_iou(bbox_a, bbox_b):
    """(len(a), len(b)) IoU matrix, no +1 terms.

    Same fp32 operation sequence as chainercv's broadcasting form
    (``prod(br - tl, axis=2) * (tl < br).all(axis=2) / (area_a + area_b - inter)``),
    written per coordinate so that no (A, B, 2) temporaries are reduced."""
    if bbox_a.shape[1] != 4 or bbox_b.shape[1] != 4:
        raise IndexError
    a = np.asarray(bbox_a, np.float32)
    b = np.asarray(bbox_b, np.float32)
    ay1, ax1, ay2, ax2 = (a[:, i, None] for i in range(4))
    by1, bx1, by2, bx2 = (b[None, :, i] for i in range(4))
    tl_y, tl_x = np.maximum(ay1, by1), np.maximum(ax1, bx1)
    br_y, br_x = np.minimum(ay2, by2), np.minimum(ax2, bx2)
    inter = (br_y - tl_y) * (br_x - tl_x)
    inter *= ((tl_y < br_y) & (tl_x < br_x))
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    return inter / (area_a[:, None] + area_b[None, :] - inter)


def bbox_iou_t(bbox_a, bbox_b):
    """``bbox_iou(a, b).T`` — the (len(b), len(a)) layout keeps the long axis contiguous, which
    is what the few-ground-truths x thousands-of-candidates calls of the target creators want
    (same fp32 operations element by element, ~4x faster than the (A, B) broadcast)."""
    a = np.ascontiguousarray(np.asarray(bbox_a, np.float32).T)     # (4, A)
    b = np.asarray(bbox_b, np.float32)
    by1, bx1, by2, bx2 = (b[:, i, None] for i in range(4))
    tl_y, tl_x = np.maximum(a[0], by1), np.maximum(a[1], bx1)
    br_y, br_x = np.minimum(a[2], by2), np.minimum(a[3], bx2)
    inter = (br_y - tl_y) * (br_x - tl_x)
    inter *= ((tl_y < br_y) & (tl_x < br_x))
    area_a = (a[2] - a[0]) * (a[3] - a[1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    return inter / (area_a[None, :] + area_b[:, None] - inter)


def bbox2loc(src_bbox, dst_bbox):
    """Offsets/scales (dy, dx, dh, dw) that map src boxes onto dst boxes."""
    src = np.asarray(src_bbox, np.float32)
    dst = np.asarray(dst_bbox, np.float32)
    half = np.float32(0.5)
    sh, sw = src[:, 2] - src[:, 0], src[:, 3] - src[:, 1]
    scy, scx = src[:, 0] + half * sh, src[:, 1] + half * sw
    dh_, dw_ = dst[:, 2] - dst[:, 0], dst[:, 3] - dst[:, 1]
    dcy, dcx = dst[:, 0] + half * dh_, dst[:, 1] + half * dw_
    eps = np.finfo(np.float32).eps
    sh, sw = np.maximum(sh, eps), np.maximum(sw, eps)
    out = np.empty((len(src), 4), np.float32)
    out[:, 0] = (dcy - scy) / sh
    out[:, 1] = (dcx - scx) / sw
    out[:, 2] = np.log(dh_ / sh)
    out[:, 3] = np.log(dw_ / sw)
    return out


def resize_bilinear(img, out_h, out_w):
    """OpenCV ``cv2.resize(img, (out_w, out_h))`` INTER_LINEAR rule for a float32
    2-D image: src = (dst + 0.5) * scale - 0.5, clamped to the border (cv2 is what
    the reference uses at models/utils/proposal_target_creator.py:171-172; it is
    not installable here)."""
    img = np.asarray(img, np.float32)
    h, w = img.shape[:2]

    def axis(n_out, n_in):
        pos = (np.arange(n_out, dtype=np.float64) + 0.5) * (n_in / float(n_out)) - 0.5
        i0 = np.floor(pos).astype(np.int64)
        t = (pos - i0).astype(np.float32)
        below, above = i0 < 0, i0 >= n_in - 1
        i0 = np.clip(i0, 0, n_in - 1)
        t[below | above] = 0.
        return i0, np.minimum(i0 + 1, n_in - 1), t

    y0, y1, ty = axis(out_h, h)
    x0, x1, tx = axis(out_w, w)
    ty = ty[:, None]
    tx = tx[None, :]
    r0, r1 = img[y0], img[y1]
    top = r0[:, x0] * (1 - tx) + r0[:, x1] * tx
    bot = r1[:, x0] * (1 - tx) + r1[:, x1] * tx
    return (top * (1 - ty) + bot * ty).astype(np.float32)
