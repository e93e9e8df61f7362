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

    # ------------------------------------------------------------------ device half (8f-3)
    def prepare_device(self, bbox, anchor_d, anchor_h, img_size, upload=None):
        """``prepare`` on the device: anchor_d (S,4) device tensor (anchor_h its host copy),
        bbox (G,4) host array.  IoU matrix, row / column maxima and the label rule run as HIP
        kernels; the labels of the inside anchors are copied to the host asynchronously (the
        caller's next synchronisation completes the copy) for the np.random draws of
        ``finish_device``."""
        import torch
        from ...functions import target_ops as T
        dev = anchor_d.device
        if upload is None:
            upload = lambda a, dt, d: torch.tensor(a, dtype=dt, device=d)
        H, W = img_size
        key = (int(H), int(W), anchor_h.shape[0], anchor_d.data_ptr())
        cache = getattr(self, '_inside_cache', None)
        if cache is None or cache[0] != key:
            inside = np.where((anchor_h[:, 0] >= 0) & (anchor_h[:, 1] >= 0) &
                              (anchor_h[:, 2] <= H) & (anchor_h[:, 3] <= W))[0].astype(np.int32)
            inside_d = torch.tensor(inside, device=dev)
            cache = (key, inside_d, anchor_d.index_select(0, inside_d.long()).contiguous())
            self._inside_cache = cache
        _, inside_d, a_d = cache
        bbox_d = upload(np.asarray(bbox, np.float32), torch.float32, dev)
        max_iou, argmax, iou, gt_max = T.bbox_iou_argmax(a_d, bbox_d, want_matrix=True)
        label_d = T.anchor_labels(iou, max_iou, gt_max, self.neg_iou_thresh, self.pos_iou_thresh)
        label_h = torch.empty(label_d.shape, dtype=torch.int32, pin_memory=True)
        label_h.copy_(label_d, non_blocking=True)
        done = torch.cuda.Event()
        done.record()
        return dict(n_anchor=int(anchor_h.shape[0]), inside=inside_d, anchor=a_d, bbox=bbox_d,
                    argmax=argmax, label_d=label_d, label_h=label_h, done=done, upload=upload)

    def finish_device(self, st):
        """The np.random draws of ``finish`` on the host copy of the labels (same calls, same
        order), then the full-size targets on the device: loc (S,4), label (S,)."""
        import torch
        from ...functions import target_ops as T
        st['done'].synchronize()
        label = st['label_h'].numpy()
        disabled = []
        n_pos = int(self.pos_ratio * self.n_sample)
        pos = np.where(label == 1)[0]
        if len(pos) > n_pos:
            d = np.random.choice(pos, size=len(pos) - n_pos, replace=False)
            label[d] = -1
            disabled.append(d)
        n_neg = self.n_sample - np.sum(label == 1)
        neg = np.where(label == 0)[0]
        if len(neg) > n_neg:
            d = np.random.choice(neg, size=len(neg) - n_neg, replace=False)
            label[d] = -1
            disabled.append(d)
        dis_d = None
        if disabled:
            dis_d = st['upload'](np.concatenate(disabled).astype(np.int32), torch.int32,
                                 st['anchor'].device)
        return T.anchor_targets_finish(st['anchor'], st['inside'], st['label_d'], st['argmax'],
                                       st['bbox'], dis_d, st['n_anchor'])

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
