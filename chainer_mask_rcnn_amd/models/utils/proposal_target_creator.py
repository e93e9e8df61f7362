"""ProposalTargetCreator — same interface and sampling semantics as the reference's
/root/reference/chainer_mask_rcnn/models/utils/proposal_target_creator.py:23-184.

Host-side NumPy, exactly like the reference (which always moves its inputs to the
CPU, :112-115): integer/sampling work whose results depend on the *global*
``np.random`` stream (seeded at examples/train_common.py:135-136), so the order of
``np.random.choice`` calls is kept: foreground first, then background.
"""
import numpy as np

from ...utils.bbox import bbox_iou_t, bbox2loc, resize_bilinear


class ProposalTargetCreator(object):
    """Assign ground truth boxes, labels and 14x14 masks to sampled RoIs."""

    def __init__(self, n_sample=512, pos_ratio=0.25, pos_iou_thresh=0.5,
                 neg_iou_thresh_hi=0.5, neg_iou_thresh_lo=0.0,
                 mask_size=14, binary_thresh=0.4):
        self.n_sample = n_sample
        self.pos_ratio = pos_ratio
        self.pos_iou_thresh = pos_iou_thresh
        self.neg_iou_thresh_hi = neg_iou_thresh_hi
        self.neg_iou_thresh_lo = neg_iou_thresh_lo
        self.mask_size = mask_size
        self.binary_thresh = binary_thresh

    def __call__(self, roi, bbox, label, mask,
                 loc_normalize_mean=(0., 0., 0., 0.),
                 loc_normalize_std=(0.1, 0.1, 0.2, 0.2)):
        """roi (R,4), bbox (G,4), label (G,), mask (G,H,W) host arrays ->
        sample_roi (S,4) f32, gt_roi_loc (S,4) f32, gt_roi_label (S,) i32 (0 = bg),
        gt_roi_mask (S,14,14) i32 in {-1,0,1}."""
        sample_roi, gt_roi_loc, gt_roi_label, job = self.sample(
            roi, bbox, label, loc_normalize_mean, loc_normalize_std)
        return sample_roi, gt_roi_loc, gt_roi_label, self.mask_targets(job, mask)

    def sample(self, roi, bbox, label, loc_normalize_mean=(0., 0., 0., 0.),
               loc_normalize_std=(0.1, 0.1, 0.2, 0.2)):
        """Everything that consumes ``np.random`` (and everything the RoI head needs before it
        can be launched): the sampled RoIs, their regression targets and labels, plus an
        opaque job for ``mask_targets``.  The train chain launches the head on the GPU right
        after this and builds the 14x14 mask targets on the host meanwhile."""
        roi = np.asarray(roi, np.float32)
        bbox = np.asarray(bbox, np.float32)
        label = np.asarray(label)
        if bbox.shape[0] == 0:
            raise ValueError('Empty bbox is not supported.')

        # ground-truth boxes are candidates too (:121)
        cand = np.concatenate((roi, bbox), axis=0)
        n_pos_max = np.round(self.n_sample * self.pos_ratio)
        iou_t = bbox_iou_t(cand, bbox)            # (G, R): the long axis contiguous
        assigned = iou_t.argmax(axis=0)
        best = iou_t.max(axis=0)
        cand_label = label[assigned] + 1          # 0 is background (:129)

        fg = np.where(best >= self.pos_iou_thresh)[0]
        n_fg = int(min(n_pos_max, fg.size))
        if fg.size > 0:
            fg = np.random.choice(fg, size=n_fg, replace=False)

        bg = np.where((best < self.neg_iou_thresh_hi) & (best >= self.neg_iou_thresh_lo))[0]
        n_bg = int(min(self.n_sample - n_fg, bg.size))
        if bg.size > 0:
            bg = np.random.choice(bg, size=n_bg, replace=False)

        chosen = np.append(fg, bg)
        gt_roi_label = cand_label[chosen].astype(np.int32)
        gt_roi_label[n_fg:] = 0
        sample_roi = cand[chosen]

        gt_roi_loc = bbox2loc(sample_roi, bbox[assigned[chosen]])
        gt_roi_loc = ((gt_roi_loc - np.array(loc_normalize_mean, np.float32)) /
                      np.array(loc_normalize_std, np.float32)).astype(np.float32)

        job = (len(sample_roi), n_fg, np.round(sample_roi[:n_fg]).astype(np.int32), assigned[fg])
        return sample_roi, gt_roi_loc, gt_roi_label, job

    # ------------------------------------------------------------------ device half (8f-3)
    def sample_device(self, roi, bbox, label, loc_normalize_mean=(0., 0., 0., 0.),
                      loc_normalize_std=(0.1, 0.1, 0.2, 0.2), upload=None):
        """``sample`` with the arithmetic on the device: roi (R,4) is a
        DEVICE tensor, bbox / label host arrays.  The IoU matrix, its row max / argmax, the
        gather of the chosen candidates, labels and normalised regression targets run as HIP
        kernels; only the per-candidate max IoU comes back to the host, where the SAME
        ``np.random.choice`` calls as ``sample`` pick the rows (foreground first, then
        background).  Returns device tensors sample_roi (S,4), gt_roi_loc (S,4), gt_roi_label
        (S,) and a job for ``mask_targets_device``; ``job['n_fg']`` / ``job['n']`` are host ints."""
        import torch
        from ...functions import target_ops as T
        bbox = np.asarray(bbox, np.float32)
        label = np.asarray(label)
        if bbox.shape[0] == 0:
            raise ValueError('Empty bbox is not supported.')
        dev = roi.device
        if upload is None:
            upload = lambda a, dt, d: torch.tensor(a, dtype=dt, device=d)
        bbox_d = upload(bbox, torch.float32, dev)
        label_d = upload(label.astype(np.int32), torch.int32, dev)
        cand = torch.cat((roi.detach().to(torch.float32), bbox_d), dim=0).contiguous()   # :121
        best_d, assigned_d = T.bbox_iou_argmax(cand, bbox_d)
        best = best_d.cpu().numpy()                    # the one read-back: (R+G,) floats
        n_pos_max = np.round(self.n_sample * self.pos_ratio)
        fg = np.where(best >= self.pos_iou_thresh)[0]
        n_fg = int(min(n_pos_max, fg.size))
        if fg.size > 0:
            fg = np.random.choice(fg, size=n_fg, replace=False)
        bg = np.where((best < self.neg_iou_thresh_hi) & (best >= self.neg_iou_thresh_lo))[0]
        n_bg = int(min(self.n_sample - n_fg, bg.size))
        if bg.size > 0:
            bg = np.random.choice(bg, size=n_bg, replace=False)
        chosen = np.append(fg, bg).astype(np.int32)
        chosen_d = upload(chosen, torch.int32, dev)
        sample_roi, gt_roi_loc, gt_roi_label, gt_index = T.proposal_targets_gather(
            cand, bbox_d, label_d, assigned_d, chosen_d, n_fg, loc_normalize_mean, loc_normalize_std)
        job = dict(n=len(chosen), n_fg=n_fg, sample_roi=sample_roi, gt_index=gt_index)
        return sample_roi, gt_roi_loc, gt_roi_label, job

    def mask_targets_device(self, job, mask):
        """14x14 mask targets on the device.  ``mask`` (G,H,W): a device tensor (uint8 / bool /
        int32) stays on the device; a host array is uploaded as uint8."""
        import torch
        from ...functions import target_ops as T
        dev = job['sample_roi'].device
        if isinstance(mask, torch.Tensor):
            m = mask.to(device=dev)
            m = (m != 0).to(torch.uint8) if m.dtype != torch.uint8 else m
        else:
            m = torch.tensor(np.ascontiguousarray(np.asarray(mask) != 0).view(np.uint8), device=dev)
        return T.mask_targets(m.contiguous(), job['sample_roi'], job['gt_index'], job['n_fg'],
                              self.mask_size)

    def mask_targets(self, job, mask):
        """14x14 mask targets for the foreground RoIs; background rows stay -1 (:160-177).
        The reference one-hot encodes the {0,1} crop, resizes both channels with bilinear
        weights (which sum to 1) and takes the argmax, i.e. foreground probability > 0.5.

        Deviation (documented): a foreground RoI whose rounded box is empty (zero height or
        width after ``np.round``) makes the reference raise inside ``gt_roi_mask_i.max()`` /
        ``cv2.resize`` on an empty crop; here its mask target is all background (0).  Such a
        RoI needs IoU >= 0.5 with a ground-truth box while being < 0.5 px wide — it cannot
        occur with COCO boxes (>= 1 px) and is covered by
        tests/test_targets_cpu.py::test_degenerate_crop_is_all_background."""
        n, n_fg, boxes, gt_index = job
        M = self.mask_size
        gt_roi_mask = -np.ones((n, M, M), dtype=np.int32)
        if n_fg > 0:
            gt_roi_mask[:n_fg] = _mask_targets(boxes, gt_index, np.asarray(mask), M)
        return gt_roi_mask


def _mask_targets(boxes, gt_index, mask, M):
    """(F, M, M) int32 mask targets for F integer boxes (y0, x0, y1, x1): bilinear resize
    (cv2 INTER_LINEAR rule, utils.bbox.resize_bilinear) of each crop ``mask[g][y0:y1, x0:x1]``
    to M x M, thresholded at 0.5 — all F crops in one vectorised gather."""
    H, W = mask.shape[1:]
    y0 = np.clip(boxes[:, 0], 0, H); y1 = np.clip(boxes[:, 2], 0, H)
    x0 = np.clip(boxes[:, 1], 0, W); x1 = np.clip(boxes[:, 3], 0, W)
    # Python slicing semantics of mask[y0:y1, x0:x1] for in-image integer boxes
    h = np.maximum(y1 - y0, 0)
    w = np.maximum(x1 - x0, 0)
    ok = (h > 0) & (w > 0)

    def axis(n_in, start):
        n = np.maximum(n_in, 1).astype(np.float64)[:, None]
        pos = (np.arange(M, dtype=np.float64)[None, :] + 0.5) * (n / float(M)) - 0.5
        i0 = np.floor(pos).astype(np.int64)
        t = (pos - i0).astype(np.float32)
        edge = (i0 < 0) | (i0 >= n.astype(np.int64) - 1)
        i0 = np.clip(i0, 0, n.astype(np.int64) - 1)
        t[edge] = 0.
        i1 = np.minimum(i0 + 1, n.astype(np.int64) - 1)
        return i0 + start[:, None], i1 + start[:, None], t

    ya, yb, ty = axis(h, y0)
    xa, xb, tx = axis(w, x0)
    ya, yb = np.clip(ya, 0, H - 1), np.clip(yb, 0, H - 1)
    xa, xb = np.clip(xa, 0, W - 1), np.clip(xb, 0, W - 1)
    g = np.asarray(gt_index)[:, None, None]
    fgm = lambda yy, xx: (mask[g, yy[:, :, None], xx[:, None, :]] > 0).astype(np.float32)
    tx_ = tx[:, None, :]
    ty_ = ty[:, :, None]
    top = fgm(ya, xa) * (1 - tx_) + fgm(ya, xb) * tx_
    bot = fgm(yb, xa) * (1 - tx_) + fgm(yb, xb) * tx_
    prob = top * (1 - ty_) + bot * ty_
    out = (prob > 0.5).astype(np.int32)
    out[~ok] = 0
    return out
