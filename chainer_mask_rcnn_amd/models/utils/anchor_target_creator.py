"""AnchorTargetCreator — chainercv's RPN target assignment (an un-vendored dependency
of the reference; default instance /root/reference/chainer_mask_rcnn/models/mask_rcnn_train_chain.py:61,
call :153-158; algorithm SURVEY.md Appendix A.5).  Host-side NumPy like upstream;
draws from the global ``np.random`` stream: positives first, then negatives."""
import numpy as np

from ...utils.bbox import bbox_iou, bbox2loc


class AnchorTargetCreator(object):

    def __init__(self, n_sample=256, pos_iou_thresh=0.7, neg_iou_thresh=0.3, pos_ratio=0.5):
        self.n_sample = n_sample
        self.pos_iou_thresh = pos_iou_thresh
        self.neg_iou_thresh = neg_iou_thresh
        self.pos_ratio = pos_ratio

    def __call__(self, bbox, anchor, img_size):
        """bbox (G,4), anchor (S,4) host arrays, img_size (H,W) ->
        loc (S,4) f32 (0 outside), label (S,) i32 in {1,0,-1}."""
        return self.finish(self.prepare(bbox, anchor, img_size))

    def prepare(self, bbox, anchor, img_size):
        """The deterministic part (IoU matrix, label rules, regression targets).  It needs
        only the ground truth, so the train chain runs it on a worker thread while the GPU
        is busy with the extractor; ``finish`` then draws from ``np.random`` on the caller's
        thread, keeping the reference's global RNG order."""
        bbox = np.asarray(bbox, np.float32)
        anchor = np.asarray(anchor, np.float32)
        H, W = img_size
        n_anchor = len(anchor)
        inside = np.where((anchor[:, 0] >= 0) & (anchor[:, 1] >= 0) &
                          (anchor[:, 2] <= H) & (anchor[:, 3] <= W))[0]
        a = anchor[inside]
        ious = bbox_iou(a, bbox)
        argmax = ious.argmax(axis=1)
        max_iou = ious[np.arange(len(a)), argmax]
        gt_best = ious.max(axis=0)
        gt_argmax = np.where(ious == gt_best[None, :])[0]     # all ties

        label = np.full((len(a),), -1, dtype=np.int32)
        label[max_iou < self.neg_iou_thresh] = 0
        label[gt_argmax] = 1
        label[max_iou >= self.pos_iou_thresh] = 1
        loc = bbox2loc(a, bbox[argmax])
        return n_anchor, inside, label, loc

    def finish(self, state):
        n_anchor, inside, label, loc = state
        n_pos = int(self.pos_ratio * self.n_sample)
        pos = np.where(label == 1)[0]
        if len(pos) > n_pos:
            label[np.random.choice(pos, size=len(pos) - n_pos, replace=False)] = -1
        n_neg = self.n_sample - np.sum(label == 1)
        neg = np.where(label == 0)[0]
        if len(neg) > n_neg:
            label[np.random.choice(neg, size=len(neg) - n_neg, replace=False)] = -1

        full_label = np.full((n_anchor,), -1, dtype=np.int32)
        full_label[inside] = label
        full_loc = np.zeros((n_anchor, 4), dtype=np.float32)
        full_loc[inside] = loc
        return full_loc, full_label
