"""ORACLE (test infrastructure, not product code) — inference post-processing.

NumPy restatement of `MaskRCNN._to_bboxes` / `_suppress`
(/root/reference/chainer_mask_rcnn/models/mask_rcnn.py:178-265) for one image, built on
np_ref's loc2bbox / non_maximum_suppression.  "parity unpinned": chainer/chainercv cannot be
imported here, and the reference has no golden vectors for this path.
"""
import numpy as np

from . import np_ref


def decode_cls_boxes(roi, roi_cls_loc, n_class, scale, size,
                     mean=(0., 0., 0., 0.), std=(0.1, 0.1, 0.2, 0.2)):
    """mask_rcnn.py:220-240: roi / scale, de-normalise, loc2bbox per class, clip."""
    roi = (roi / np.float32(scale)).astype(np.float32)
    mean = np.tile(np.asarray(mean, np.float32), n_class)
    std = np.tile(np.asarray(std, np.float32), n_class)
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
