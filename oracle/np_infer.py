"""ORACLE (test infrastructure, not product code) — inference post-processing.

NumPy restatement of `MaskRCNN._to_bboxes` / `_suppress`
(/root/reference/chainer_mask_rcnn/models/mask_rcnn.py:178-265) for one image, built on
np_ref's loc2bbox / non_maximum_suppression.  The control flow is pinned to the reference's own
method bodies by tests/golden/to_bboxes.npz (oracle/gen_golden.py section 7); the chainercv
helpers underneath (loc2bbox, NMS) remain "parity unpinned": chainercv cannot be imported here.
"""
import numpy as np

from . import np_ref


def decode_cls_boxes(roi, roi_cls_loc, n_class, scale, size,
                     mean=(0., 0., 0., 0.), std=(0.1, 0.1, 0.2, 0.2)):
    """mask_rcnn.py:220-240: roi / scale, de-normalise, loc2bbox per class, clip."""
    roi = (roi / np.float32(scale)).astype(np.float32)
    # tuples of Python floats -> float64 arrays, as `xp.asarray(self.loc_normalize_mean)` gives:
    # the de-normalisation runs in double and is rounded once by the astype
    mean = np.tile(np.asarray(mean), n_class)
    std = np.tile(np.asarray(std), n_class)
    loc = (roi_cls_loc * std + mean).astype(np.float32).reshape((-1, n_class, 4))
    roi_cls = np.broadcast_to(roi[:, None], loc.shape)
    cls_bbox = np_ref.loc2bbox(roi_cls.reshape((-1, 4)), loc.reshape((-1, 4)))
    cls_bbox = cls_bbox.reshape((-1, n_class * 4))
    cls_bbox[:, 0::2] = np.clip(cls_bbox[:, 0::2], 0, size[0])
    cls_bbox[:, 1::2] = np.clip(cls_bbox[:, 1::2], 0, size[1])
    return cls_bbox


def suppress(raw_cls_bbox, raw_prob, n_class, nms_thresh=0.5, score_thresh=0.05):
    """mask_rcnn.py:178-202."""
    bbox, label, score = [], [], []
    for l in range(1, n_class):
        cls_bbox_l = raw_cls_bbox.reshape((-1, n_class, 4))[:, l, :]
        prob_l = raw_prob[:, l]
        keep = prob_l > score_thresh
        cls_bbox_l = cls_bbox_l[keep]
        prob_l = prob_l[keep]
        keep = np_ref.non_maximum_suppression(cls_bbox_l, nms_thresh, prob_l)
        bbox.append(cls_bbox_l[keep])
        label.append((l - 1) * np.ones((len(keep),)))
        score.append(prob_l[keep])
    bbox = np.concatenate(bbox, axis=0).astype(np.float32)
    label = np.concatenate(label, axis=0).astype(np.int32)
    score = np.concatenate(score, axis=0).astype(np.float32)
    return bbox, label, score


def finish(bbox, label, score, detections_per_im=100):
    """mask_rcnn.py:247-260, including the argsort-vs-rank expression as written."""
    bbox_int = np.round(bbox).astype(np.int32)
    sizes = (bbox_int[:, 2] - bbox_int[:, 0]) * (bbox_int[:, 3] - bbox_int[:, 1])
    keep = sizes > 0
    bbox, label, score = bbox[keep], label[keep], score[keep]
    if detections_per_im > 0:
        indices = np.argsort(score, kind='stable')
        keep = indices >= (len(indices) - detections_per_im)
        bbox, label, score = bbox[keep], label[keep], score[keep]
    return bbox, label, score


# --------------------------------------------------------------------------------------
# cv2-free restatement of the image I/O of MaskRCNN.predict (mask_rcnn.py:44-107,152-176).
# OpenCV's resize(INTER_LINEAR) for float32 images: for every destination index d
#   f = (float)((d + 0.5) * scale - 0.5); s = floor(f); f -= s;  clamp s to [0, n-1] with f = 0,
# horizontal blend first, then vertical.  cv2 is not installable here: "parity unpinned".
# --------------------------------------------------------------------------------------

def _cv_axis(n_out, n_in, scale):
    d = np.arange(n_out, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    t = (f - s.astype(np.float32)).astype(np.float32)
    lo, hi = s < 0, s >= n_in - 1
    s = np.clip(s, 0, n_in - 1)
    t[lo | hi] = 0.
    return s, np.minimum(s + 1, n_in - 1), t


def cv_resize_linear(img, out_h, out_w, scale_y=None, scale_x=None):
    """img (h, w) float32 -> (out_h, out_w); scale_* = source pixels per destination pixel
    (default n_in / n_out, what cv2.resize(img, (w, h)) uses; cv2.resize(img, None, fx=s)
    uses 1 / s)."""
    img = np.asarray(img, np.float32)
    h, w = img.shape
    y0, y1, ty = _cv_axis(out_h, h, h / float(out_h) if scale_y is None else scale_y)
    x0, x1, tx = _cv_axis(out_w, w, w / float(out_w) if scale_x is None else scale_x)
    one = np.float32(1)
    r0, r1 = img[y0], img[y1]
    top = r0[:, x0] * (one - tx)[None, :] + r0[:, x1] * tx[None, :]
    bot = r1[:, x0] * (one - tx)[None, :] + r1[:, x1] * tx[None, :]
    return (top * (one - ty)[:, None] + bot * ty[:, None]).astype(np.float32)


def _cv_round_short(v):
    """saturate_cast<short>(float): cvRound = round half to even."""
    return np.clip(np.rint(np.asarray(v, np.float32)), -32768, 32767).astype(np.int32)


def cv_resize_linear_u8(img, out_h, out_w, scale_y=None, scale_x=None):
    """cv2.resize(INTER_LINEAR) of an (h, w) uint8 image: OpenCV's 8-bit FIXED-POINT path
    (imgproc/resize.cpp, resizeGeneric_ + HResizeLinear<uchar,int,short> +
    VResizeLinear<uchar,int,short,FixedPtCast<.., 22>>), restated from the published source:

      horizontal: fx as in the float path (clamped to the border with fx = 0),
                  a = saturate_cast<short>({1 - fx, fx} * 2048),   D = S[sx]*a0 + S[sx+1]*a1   (int)
      vertical:   fy = (float)((dy + .5)*scale - .5) - floor(..)  (NOT clamped), the two source
                  rows clipped to [0, h-1],  b = saturate_cast<short>({1 - fy, fy} * 2048),
                  dst = uchar((((b0 * (D0 >> 4)) >> 16) + ((b1 * (D1 >> 4)) >> 16) + 2) >> 2)

    (the exact 2x down-scale, which OpenCV reroutes to INTER_AREA, is not special-cased: the
    min_size / max_size rule of MaskRCNN.prepare never scales by exactly 1/2 on COCO sizes up).
    cv2 is not installable here: "parity unpinned"."""
    img = np.asarray(img)
    assert img.dtype == np.uint8 and img.ndim == 2
    h, w = img.shape
    x0, x1, tx = _cv_axis(out_w, w, w / float(out_w) if scale_x is None else scale_x)
    a0, a1 = _cv_round_short((np.float32(1) - tx) * np.float32(2048)), _cv_round_short(tx * np.float32(2048))
    sy = h / float(out_h) if scale_y is None else scale_y
    d = np.arange(out_h, dtype=np.float64)
    fy = ((d + 0.5) * sy - 0.5).astype(np.float32)
    s = np.floor(fy).astype(np.int64)
    fy = (fy - s.astype(np.float32)).astype(np.float32)
    y0, y1 = np.clip(s, 0, h - 1), np.clip(s + 1, 0, h - 1)
    b0, b1 = _cv_round_short((np.float32(1) - fy) * np.float32(2048)), _cv_round_short(fy * np.float32(2048))
    src = img.astype(np.int32)
    D = src[:, x0] * a0[None, :] + src[:, x1] * a1[None, :]          # (h, out_w) int32
    D0, D1 = D[y0] >> 4, D[y1] >> 4
    out = (((b0[:, None] * D0) >> 16) + ((b1[:, None] * D1) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def prepare(img, mean, min_size, max_size):
    """MaskRCNN.prepare for one CHW image (mask_rcnn.py:152-176) -> (prepared CHW f32, scale).
    uint8 images take OpenCV's 8-bit fixed-point resize (result rounded to uint8 before the
    mean is subtracted), float images the float path — as cv2.resize dispatches on depth."""
    _, H, W = img.shape
    scale = 1.
    if min_size:
        scale = min_size / min(H, W)
    if max_size and scale * max(H, W) > max_size:
        scale = max_size / max(H, W)
    out_h, out_w = int(np.round(H * scale)), int(np.round(W * scale))   # cvRound(src * f)
    if np.asarray(img).dtype == np.uint8:
        out = np.stack([cv_resize_linear_u8(img[c], out_h, out_w, 1. / scale, 1. / scale)
                        for c in range(img.shape[0])]).astype(np.float32)
    else:
        out = np.stack([cv_resize_linear(img[c].astype(np.float32), out_h, out_w, 1. / scale, 1. / scale)
                        for c in range(img.shape[0])])
    return (out - np.asarray(mean, np.float32).reshape(-1, 1, 1)).astype(np.float32), scale


def expand_boxes(boxes, scale):
    """mask_rcnn.py:44-60 (float32 arithmetic on float32 boxes, float64 result array)."""
    w_half = (boxes[:, 2] - boxes[:, 0]) * np.float32(.5)
    h_half = (boxes[:, 3] - boxes[:, 1]) * np.float32(.5)
    x_c = (boxes[:, 2] + boxes[:, 0]) * np.float32(.5)
    y_c = (boxes[:, 3] + boxes[:, 1]) * np.float32(.5)
    w_half = w_half * np.float32(scale)
    h_half = h_half * np.float32(scale)
    out = np.zeros(boxes.shape)
    out[:, 0] = x_c - w_half
    out[:, 2] = x_c + w_half
    out[:, 1] = y_c - h_half
    out[:, 3] = y_c + h_half
    return out


def segm_results(bbox, label, roi_mask_logits, im_h, im_w):
    """mask_rcnn.py:63-107 with the sigmoid of _to_masks (:296) folded in.
    bbox (D,4) yx f32, label (D,), roi_mask_logits (D, n_fg, M, M) -> (D, im_h, im_w) bool."""
    if len(bbox) == 0:
        return np.zeros((0, im_h, im_w), dtype=bool)
    M = roi_mask_logits.shape[2]
    prob = (1. / (1. + np.exp(-roi_mask_logits.astype(np.float64)))).astype(np.float32)
    ref_boxes = expand_boxes(bbox[:, [1, 0, 3, 2]].astype(np.float32), (M + 2.0) / M)
    ref_boxes = ref_boxes.astype(np.int32)
    padded = np.zeros((M + 2, M + 2), dtype=np.float32)
    out = []
    for i in range(len(ref_boxes)):
        padded[1:-1, 1:-1] = prob[i, label[i]]
        rb = ref_boxes[i]
        w = max(rb[2] - rb[0] + 1, 1)
        h = max(rb[3] - rb[1] + 1, 1)
        mask = (cv_resize_linear(padded, h, w) > 0.5).astype(np.uint8)
        im_mask = np.zeros((im_h, im_w), dtype=np.uint8)
        x_0, x_1 = max(rb[0], 0), min(rb[2] + 1, im_w)
        y_0, y_1 = max(rb[1], 0), min(rb[3] + 1, im_h)
        if x_1 > x_0 and y_1 > y_0:
            im_mask[y_0:y_1, x_0:x_1] = mask[(y_0 - rb[1]):(y_1 - rb[1]), (x_0 - rb[0]):(x_1 - rb[0])]
        out.append(im_mask.astype(bool))
    return np.asarray(out)
