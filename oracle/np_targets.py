"""ORACLE (test infrastructure, not product code) — ProposalTargetCreator.

Restatement of /root/reference/chainer_mask_rcnn/models/utils/proposal_target_creator.py:63-184
with the same statement order (so the global np.random stream is consumed identically),
including the one-hot -> resize -> argmax mask-target construction (:164-177).  `cv2.resize`
is not installable here; np_ref.resize_bilinear restates OpenCV's INTER_LINEAR rule, so the
resize underneath is "parity unpinned"; the class as a whole is pinned to the reference's own
class body by tests/golden/proposal_target_creator.npz (oracle/gen_golden.py section 6).
"""
import numpy as np

from . import np_ref


class ProposalTargetCreator(object):

    def __init__(self, n_sample=512, pos_ratio=0.25, pos_iou_thresh=0.5,
                 neg_iou_thresh_hi=0.5, neg_iou_thresh_lo=0.0, mask_size=14,
                 binary_thresh=0.4):
        self.n_sample = n_sample
        self.pos_ratio = pos_ratio
        self.pos_iou_thresh = pos_iou_thresh
        self.neg_iou_thresh_hi = neg_iou_thresh_hi
        self.neg_iou_thresh_lo = neg_iou_thresh_lo
        self.mask_size = mask_size
        self.binary_thresh = binary_thresh

    def __call__(self, roi, bbox, label, mask, loc_normalize_mean=(0., 0., 0., 0.),
                 loc_normalize_std=(0.1, 0.1, 0.2, 0.2)):
        n_bbox, _ = bbox.shape
        if n_bbox == 0:
            raise ValueError('Empty bbox is not supported.')
        roi = np.concatenate((roi, bbox), axis=0)                                   # :121
        pos_roi_per_image = np.round(self.n_sample * self.pos_ratio)
        iou = np_ref.bbox_iou(roi, bbox)
        gt_assignment = iou.argmax(axis=1)
        max_iou = iou.max(axis=1)
        gt_roi_label = label[gt_assignment] + 1                                     # :129
        pos_index = np.where(max_iou >= self.pos_iou_thresh)[0]
        pos_roi_per_this_image = int(min(pos_roi_per_image, pos_index.size))
        if pos_index.size > 0:
            pos_index = np.random.choice(pos_index, size=pos_roi_per_this_image, replace=False)
        neg_index = np.where((max_iou < self.neg_iou_thresh_hi) &
                             (max_iou >= self.neg_iou_thresh_lo))[0]
        neg_roi_per_this_image = self.n_sample - pos_roi_per_this_image
        neg_roi_per_this_image = int(min(neg_roi_per_this_image, neg_index.size))
        if neg_index.size > 0:
            neg_index = np.random.choice(neg_index, size=neg_roi_per_this_image, replace=False)
        keep_index = np.append(pos_index, neg_index)
        gt_roi_label = gt_roi_label[keep_index]
        gt_roi_label[pos_roi_per_this_image:] = 0
        sample_roi = roi[keep_index]
        gt_roi_loc = np_ref.bbox2loc(sample_roi, bbox[gt_assignment[keep_index]])
        gt_roi_loc = ((gt_roi_loc - np.array(loc_normalize_mean, np.float32)) /
                      np.array(loc_normalize_std, np.float32))
        gt_roi_mask = - np.ones((len(sample_roi), self.mask_size, self.mask_size), dtype=np.int32)
        for i, pos_ind in enumerate(pos_index):                                     # :164-177
            r = np.round(sample_roi[i]).astype(np.int32)
            gt_mask = mask[gt_assignment[pos_ind]]
            m = gt_mask[r[0]:r[2], r[1]:r[3]]
            if m.size == 0:
                # the reference raises on an empty crop (`m.max()` of an empty array); the
                # build returns an all-background target instead
                gt_roi_mask[i] = 0
                continue
            score = (np.arange(m.max() + 1) == m[..., None]).astype(np.float32)
            score = np.stack([np_ref.resize_bilinear(score[..., c], self.mask_size, self.mask_size)
                              for c in range(score.shape[2])], axis=2)
            gt_roi_mask[i] = np.argmax(score, axis=2).astype(np.int32)
        return sample_roi, gt_roi_loc, gt_roi_label.astype(np.int32), gt_roi_mask
