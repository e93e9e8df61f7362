"""MaskRCNN base model — same interface as the reference's
/root/reference/chainer_mask_rcnn/models/mask_rcnn.py:110-337 for the hot path
(``__call__``, ``_to_bboxes`` / ``_suppress`` / ``_to_roi_masks``, ``predict``).

Inference post-processing runs on the device: softmax, per-class decode+clip, a batched
per-class stable sort and one batched bit-mask NMS launch for all 80 classes (the
reference loops over classes in Python on CPU copies, :178-202).  ``prepare`` and
``segm_results`` (cv2 resize / paste of the final image-size masks) are host-side image
I/O and are out of the hot-path scope ("next" row f-2); ``predict``
therefore returns the 14x14 per-detection mask probabilities instead of pasted masks.
"""
import numpy as np
import torch

from .. import functions as F
from .. import _lib
from ..functions import proposal_ops as P


class MaskRCNN(torch.nn.Module):

    def __init__(self, extractor, rpn, head, mean, min_size=600, max_size=1000,
                 loc_normalize_mean=(0., 0., 0., 0.),
                 loc_normalize_std=(0.1, 0.1, 0.2, 0.2), detections_per_im=100):
        super(MaskRCNN, self).__init__()
        self.extractor = extractor
        self.rpn = rpn
        self.head = head

        self.mean = mean
        self.min_size = min_size
        self.max_size = max_size
        self.loc_normalize_mean = loc_normalize_mean
        self.loc_normalize_std = loc_normalize_std

        self.nms_thresh = 0.5
        self.score_thresh = 0.05

        self._detections_per_im = detections_per_im

    @property
    def n_class(self):
        # Total number of classes including the background.
        return self.head.n_class

    def forward(self, x, scales):
        img_size = x.shape[2:]
        h = self.extractor(x)
        rpn_locs, rpn_scores, rois, roi_indices, anchor = self.rpn(h, img_size, scales)
        roi_cls_locs, roi_scores, roi_masks = self.head(h, rois, roi_indices)
        return roi_cls_locs, roi_scores, rois, roi_indices, roi_masks

    # ------------------------------------------------------------------ inference
    def _suppress_queue(self, cls_bbox, prob):
        """Device half of the per-class score threshold + NMS (:178-202) for one image, all
        classes batched: queues the kernels and returns the device results without
        synchronising.  cls_bbox (R, n_class, 4), prob (R, n_class) device tensors."""
        R = cls_bbox.shape[0]
        n_fg = self.n_class - 1
        dev = cls_bbox.device
        sorted_boxes = torch.empty((n_fg, max(R, 1), 4), dtype=torch.float32, device=dev)
        sorted_prob = torch.empty((n_fg, max(R, 1)), dtype=torch.float32, device=dev)
        counts = torch.empty((n_fg,), dtype=torch.int32, device=dev)
        ws = _lib.workspace(_lib.load().mrcnn_detect_sort_workspace_bytes(R, self.n_class), dev,
                            'detect')
        prob, cls_bbox = prob.contiguous(), cls_bbox.contiguous()
        _lib.call('mrcnn_detect_sort', _lib.ptr(prob), _lib.ptr(cls_bbox),
                  R, self.n_class, float(self.score_thresh), _lib.ptr(sorted_boxes),
                  _lib.ptr(sorted_prob), _lib.ptr(counts), _lib.ptr(ws), _lib.stream_ptr())
        keep, n_keep = P.nms_sorted_batched(sorted_boxes, counts, self.nms_thresh)
        # pack the kept rows of all classes on the device: the host then reads one count and
        # D rows instead of four (n_fg, R) arrays
        cap = n_fg * max(R, 1)
        bbox = torch.empty((cap, 4), dtype=torch.float32, device=dev)
        label = torch.empty((cap,), dtype=torch.int32, device=dev)
        score = torch.empty((cap,), dtype=torch.float32, device=dev)
        total = torch.empty((1,), dtype=torch.int32, device=dev)
        _lib.call('mrcnn_detect_compact', _lib.ptr(keep), _lib.ptr(n_keep), _lib.ptr(sorted_boxes),
                  _lib.ptr(sorted_prob), n_fg, max(R, 1) if R else 0, _lib.ptr(bbox), _lib.ptr(label),
                  _lib.ptr(score), _lib.ptr(total), _lib.stream_ptr())
        return bbox, label, score, total

    @staticmethod
    def _suppress_finish(bbox, label, score, total):
        """Host half: read the packed rows back.  Returns host arrays bbox (D,4) f32, label (D,)
        i32, score (D,) f32 ordered by class then by score."""
        d = int(total.item())
        return (bbox[:d].cpu().numpy().astype(np.float32, copy=False),
                label[:d].cpu().numpy().astype(np.int32, copy=False),
                score[:d].cpu().numpy().astype(np.float32, copy=False))

    def _suppress(self, cls_bbox, prob):
        return self._suppress_finish(*self._suppress_queue(cls_bbox, prob))

    def _to_bboxes(self, roi_cls_locs, roi_scores, rois, roi_indices, sizes, scales):
        queued = self._queue_detections(roi_cls_locs, roi_scores, rois, roi_indices, sizes, scales)
        bboxes, labels, scores = [], [], []
        for q in queued:
            bbox, label, score = self._finish_detections(q)
            bboxes.append(bbox)
            labels.append(label)
            scores.append(score)
        return bboxes, labels, scores

    def _queue_detections(self, roi_cls_locs, roi_scores, rois, roi_indices, sizes, scales,
                          bounds=None):
        """Device half of ``_to_bboxes`` for every image, queued without synchronising
        (``bounds``: the per-image row bounds of ``rois`` if the caller already knows them —
        reading ``roi_indices`` back would wait for everything queued so far)."""
        probs = F.softmax(roi_scores.detach())
        roi_cls_locs = roi_cls_locs.detach()
        if roi_cls_locs.stride(1) != 1:
            roi_cls_locs = roi_cls_locs.contiguous()
        import ctypes
        mean = (ctypes.c_double * 4)(*[float(v) for v in self.loc_normalize_mean])
        std = (ctypes.c_double * 4)(*[float(v) for v in self.loc_normalize_std])
        # RoIs are grouped by image, in order (RegionProposalNetwork): one host read gives the
        # slice bounds.  The device work of every image is queued first; the host halves run
        # afterwards while later images' kernels are still executing.
        if bounds is None:
            counts = np.bincount(roi_indices.cpu().numpy().astype(np.int64), minlength=len(sizes))
            bounds = np.concatenate([[0], np.cumsum(counts)])
        queued = []
        for index in range(len(sizes)):
            lo, hi = int(bounds[index]), int(bounds[index + 1])
            roi = rois[lo:hi].contiguous()
            loc = roi_cls_locs[lo:hi]
            prob = probs[lo:hi].contiguous()
            R = roi.shape[0]
            cls_bbox = torch.empty((R, self.n_class, 4), dtype=torch.float32, device=roi.device)
            if loc.stride(1) != 1 or loc.stride(0) != loc.shape[1]:
                loc = loc.contiguous()
            _lib.call('mrcnn_decode_cls_boxes', _lib.ptr(roi), _lib.ptr(loc), loc.stride(0),
                      _lib.ptr(cls_bbox), R, self.n_class, float(scales[index]), mean, std,
                      float(sizes[index][0]), float(sizes[index][1]), _lib.stream_ptr())
            queued.append(self._suppress_queue(cls_bbox, prob))
        return queued

    def _finish_detections(self, q):
        """Host half of ``_to_bboxes`` for one image (:247-260)."""
        bbox, label, score = self._suppress_finish(*q)

        bbox_int = np.round(bbox).astype(np.int32)
        bbox_sizes = ((bbox_int[:, 2] - bbox_int[:, 0]) * (bbox_int[:, 3] - bbox_int[:, 1]))
        ok = bbox_sizes > 0
        bbox, label, score = bbox[ok], label[ok], score[ok]

        if self._detections_per_im > 0:
            # literal restatement of models/mask_rcnn.py:255-260 (an argsort
            # permutation compared with a rank threshold).  The reference
            # calls np.argsort with its default (unstable) kind, so the order of EQUAL scores —
            # and with it which of two tied detections survives the cut — is unspecified there;
            # the stable kind used here is one of the orders it can produce.
            indices = np.argsort(score, kind='stable')
            ok = indices >= (len(indices) - self._detections_per_im)
            bbox, label, score = bbox[ok], label[ok], score[ok]
        return bbox, label, score

    def _roi_masks_group(self, h, bboxes, image_ids, scales):
        """Mask-head pass (:279-289) for the detections of the images ``image_ids`` (one head
        launch for the group); returns one device tensor (D_i, n_fg, M, M) per image."""
        n_fg_class, mask_size = self.n_class - 1, self.head.mask_size
        counts = [len(b) for b in bboxes]
        if sum(counts) == 0:
            return [torch.zeros((0, n_fg_class, mask_size, mask_size), device=h.device)
                    for _ in bboxes]
        rois = np.concatenate([b * np.float32(scales[i]) for b, i in zip(bboxes, image_ids)], axis=0)
        idx = np.concatenate([np.full((c,), i, dtype=np.int32) for c, i in zip(counts, image_ids)])
        with torch.no_grad():
            _, _, roi_masks = self.head(h, torch.tensor(rois, dtype=torch.float32, device=h.device),
                                        torch.tensor(idx, device=h.device), pred_bbox=False)
        bounds = np.concatenate([[0], np.cumsum(counts)])
        return [roi_masks[int(bounds[k]):int(bounds[k + 1])] for k in range(len(bboxes))]

    def _to_roi_masks(self, h, bboxes, roi_indices, scales, to_host=True):
        batch_size = h.shape[0]
        bboxes = np.concatenate(bboxes, axis=0)
        n_fg_class = self.n_class - 1
        mask_size = self.head.mask_size
        if bboxes.size == 0:
            if to_host:
                return [np.zeros((0, n_fg_class, mask_size, mask_size), dtype=np.float32)
                        for _ in range(batch_size)]
            return [torch.zeros((0, n_fg_class, mask_size, mask_size), device=h.device)
                    for _ in range(batch_size)]
        with torch.no_grad():
            scales = np.asarray(scales, dtype=np.float32)
            rois = bboxes * scales[roi_indices][:, None]
            rois = torch.tensor(rois, dtype=torch.float32, device=h.device)
            _, _, roi_masks = self.head(
                h, rois, torch.tensor(roi_indices, device=h.device), pred_bbox=False)
        if not to_host:
            idx = torch.tensor(roi_indices, device=h.device)
            return [roi_masks[idx == i] for i in range(batch_size)]
        roi_masks = roi_masks.cpu().numpy()
        return [roi_masks[roi_indices == i] for i in range(batch_size)]

    # ------------------------------------------------------------------ image I/O on device
    def prepare(self, imgs, x_flips=None):
        """models/mask_rcnn.py:152-176 + concat_examples(padding=0): a list of CHW RGB images
        (uint8 or float) -> zero-padded device batch x (N,3,H,W) channels-last, original
        sizes and scales.  The bilinear resize (cv2 INTER_LINEAR rule) and the mean
        subtraction run in one HIP kernel per image; uint8 images are uploaded as they are.
        ``x_flips`` (extension, used by datasets.MaskRCNNTransform): per-image booleans, mirror
        the resized image left-right in the same pass."""
        dev = next(self.parameters()).device
        sizes, scales, outs = [], [], []
        for img in imgs:
            _, H, W = img.shape
            scale = 1.
            if self.min_size:
                scale = self.min_size / min(H, W)
            if self.max_size and scale * max(H, W) > self.max_size:
                scale = self.max_size / max(H, W)
            sizes.append((H, W))
            scales.append(scale)
            outs.append((int(np.round(H * scale)), int(np.round(W * scale))))
        N = len(imgs)
        Hm, Wm = max(o[0] for o in outs), max(o[1] for o in outs)
        batch = torch.zeros((N, Hm, Wm, 3), dtype=torch.float32, device=dev)
        mean = (_lib.c_f32 * 3)(*[float(v) for v in np.asarray(self.mean).ravel()])
        for n, img in enumerate(imgs):
            is_u8 = getattr(img, 'dtype', None) == np.uint8
            host = np.ascontiguousarray(img, dtype=np.uint8 if is_u8 else np.float32)
            if dev.type == 'cuda':
                # through pinned memory, asynchronously: a pageable hipMemcpy is staged by the
                # runtime and blocks the calling thread (measured 8-12 ms for 1 MB next to a busy
                # compute stream); torch's pinned-block cache keeps the buffer alive until the
                # copy has run
                stage = torch.empty(host.shape, dtype=torch.uint8 if is_u8 else torch.float32,
                                    pin_memory=True)
                np.copyto(stage.numpy(), host)       # (plain memcpy: no OpenMP team for 1 MB)
                src = stage.to(dev, non_blocking=True)
            else:
                src = torch.as_tensor(host).to(dev)
            flip = bool(x_flips[n]) if x_flips is not None else False
            _lib.call('mrcnn_prepare_image', _lib.ptr(src), int(is_u8), 3, sizes[n][0], sizes[n][1],
                      float(scales[n]), mean, _lib.ptr(batch), Hm, Wm, outs[n][0], outs[n][1], n,
                      int(flip), _lib.stream_ptr())
        return batch.permute(0, 3, 1, 2), sizes, scales

    def _to_masks(self, bboxes, labels, scores, roi_masks, sizes):
        """sigmoid + segm_results (models/mask_rcnn.py:63-107,292-305) on the device:
        per-image (D, im_h, im_w) bool arrays."""
        masks = []
        for bbox, label, roi_mask, size in zip(bboxes, labels, roi_masks, sizes):
            D = len(bbox)
            if D == 0:
                masks.append(np.zeros((0, size[0], size[1]), dtype=bool))
                continue
            from ..functions._layout import nhwc
            logits = nhwc(roi_mask)
            dev = logits.device
            out = torch.empty((D, size[0], size[1]), dtype=torch.uint8, device=dev)
            label_d = torch.tensor(label, dtype=torch.int32, device=dev)
            bbox_d = torch.tensor(bbox, dtype=torch.float32, device=dev)
            _lib.call('mrcnn_paste_masks', _lib.ptr(logits), _lib.ptr(label_d), _lib.ptr(bbox_d),
                      D, logits.shape[2], logits.shape[1], int(size[0]), int(size[1]),
                      _lib.ptr(out), _lib.stream_ptr())
            masks.append(out.cpu().numpy().astype(bool))
        return masks

    def predict(self, imgs):
        """Detect objects in a list of CHW RGB images (models/mask_rcnn.py:307-337):
        returns (bboxes, masks, labels, scores), per-image lists as the reference."""
        x, sizes, scales = self.prepare(imgs)
        bboxes, roi_masks, labels, scores = self.predict_prepared(
            x, scales, sizes, masks_to_host=False)
        masks = self._to_masks(bboxes, labels, scores, roi_masks, sizes)
        return bboxes, masks, labels, scores

    def predict_prepared(self, x, scales, sizes, masks_to_host=True, return_intermediates=False):
        """The device part of ``predict`` (:311-335) on an already prepared, zero-padded
        batch x (N,3,H,W) with per-image ``scales`` and original ``sizes`` (H,W).

        Returns (bboxes, roi_mask_logits, labels, scores): per-image lists of host arrays;
        roi_mask_logits[i] is (D_i, n_fg_class, 14, 14).  ``return_intermediates`` appends a
        dict with the head outputs the detections were computed from (``roi_cls_locs``,
        ``roi_scores``, ``rois``, ``roi_indices`` device tensors, ``feature_shape``)."""
        from .. import optimizers
        optimizers.flush_all()             # no parameter update may be pending while predicting
        was_training = self.training
        self.eval()
        try:
            with torch.no_grad():
                h = self.extractor(x)
                rpn_locs, rpn_scores, rois, roi_indices, anchor = self.rpn(
                    h, x.shape[2:], scales)
                # one image's proposals at a time: keeps every activation below the 2 GiB
                # buffer-addressing limit of the conv kernels (8 x 1000 RoIs would not fit)
                # (rois are grouped by image in order: one host read of the index column gives
                # the slice bounds; a boolean mask per image would synchronise eight times and
                # leave the GPU idle while the host queues the next head)
                counts = getattr(self.rpn, 'last_counts', None)   # host ints of THIS rpn call
                if counts is None or len(counts) != x.shape[0]:
                    counts = np.bincount(roi_indices.cpu().numpy(), minlength=x.shape[0])
                bounds = np.concatenate([[0], np.cumsum(counts)])
                locs, scs = [], []
                for i in range(x.shape[0]):
                    lo, hi = int(bounds[i]), int(bounds[i + 1])
                    l_i, s_i, _ = self.head(h, rois[lo:hi], roi_indices[lo:hi], pred_mask=False)
                    locs.append(l_i)
                    scs.append(s_i)
                roi_cls_locs = torch.cat(locs, dim=0)
                roi_scores = torch.cat(scs, dim=0)
            # Detections: every image's device work is queued first; the host halves then run
            # group by group and each group's mask-head pass is launched as soon as its
            # boxes are final, so the GPU computes masks while the host finishes the next
            # group (rows of the head are independent: same values as one pass over all).
            # (the row bounds are already on the host: the detection kernels queue up right
            # behind the heads instead of after a read-back that drains the GPU)
            queued = self._queue_detections(roi_cls_locs, roi_scores, rois, roi_indices,
                                            sizes, scales, bounds=bounds)
            n_img = len(queued)
            n_groups = 2 if n_img >= 4 else 1
            per = -(-n_img // n_groups)
            bboxes, labels, scores, roi_masks = [], [], [], []
            for g0 in range(0, n_img, per):
                ids = list(range(g0, min(g0 + per, n_img)))
                for i in ids:
                    b, l, s = self._finish_detections(queued[i])
                    bboxes.append(b)
                    labels.append(l)
                    scores.append(s)
                roi_masks += self._roi_masks_group(h, [bboxes[i] for i in ids], ids, scales)
            if masks_to_host:
                # (N x <=100 x 80 x 14 x 14 floats: ~50 MB for a batch of 8) through pinned
                # host memory, all copies queued before the one synchronisation
                host = [torch.empty(m.shape, dtype=m.dtype, pin_memory=True) for m in roi_masks]
                for h_, m in zip(host, roi_masks):
                    h_.copy_(m, non_blocking=True)
                torch.cuda.current_stream(x.device).synchronize()
                roi_masks = [h_.numpy() for h_ in host]
        finally:
            self.train(was_training)
        if return_intermediates:
            return bboxes, roi_masks, labels, scores, dict(
                roi_cls_locs=roi_cls_locs, roi_scores=roi_scores, rois=rois,
                roi_indices=roi_indices, feature_shape=tuple(h.shape))
        return bboxes, roi_masks, labels, scores
