"""MaskRCNNTrainChain — same interface and call sequence as the reference's
/root/reference/chainer_mask_rcnn/models/mask_rcnn_train_chain.py:25-189.

forward(imgs, bboxes, labels, masks, scales) -> scalar loss (sum of the five losses);
the six reported scalars are kept in ``self.report`` (device tensors, no sync) as the
reference reports them through chainer.reporter (:182-188).
"""
import time

import numpy as np
import torch

from .. import functions as F
from .utils import AnchorTargetCreator
from .utils import ProposalTargetCreator


_POOL = None
_COPY_STREAMS = {}


def copy_stream(dev):
    """The host-to-device copy stream of ``dev`` (one per device for the whole process: input
    pipelines should upload on it rather than create their own — the step uses four HIP streams
    (compute, copy, weight gradients, deferred work) and a GPU has four hardware queues by default;
    a fifth stream shares a queue with one of them and serialises work that was meant to overlap)."""
    key = str(dev)
    if key not in _COPY_STREAMS:
        _COPY_STREAMS[key] = torch.cuda.Stream(device=dev)
    return _COPY_STREAMS[key]


_NP_OF = {torch.float32: np.float32, torch.int32: np.int32, torch.int64: np.int64,
          torch.uint8: np.uint8}


class _PinnedRing(object):
    """Pinned host staging buffers for ``_upload``: a buffer is reused once the copy that read it
    has completed (its event), so the host never waits for a copy and never for the copy stream's
    other work (the input pipeline's next-batch upload shares that stream)."""

    MAX_BYTES = 256 << 20        # pinned bytes the ring may hold (buffers in flight included)

    def __init__(self):
        import threading
        self.lock = threading.Lock()
        self.free = []           # [(buffer, event of its last copy or None)], oldest first
        self.total = 0           # bytes of every live buffer: the free list + those handed out

    def take(self, nbytes):
        with self.lock:
            for i, (buf, ev) in enumerate(self.free):
                if buf.numel() >= nbytes and (ev is None or ev.query()):
                    return self.free.pop(i)[0]
            size = max(int(nbytes), 1 << 16)
            # at the cap: wait for the OLDEST copy and recycle / release its buffer instead of
            # growing (the copy stream is backed up; pinned memory must stay bounded)
            while self.free and self.total + size > self.MAX_BYTES:
                buf, ev = self.free.pop(0)
                if ev is not None:
                    ev.synchronize()
                if buf.numel() >= nbytes:
                    return buf
                self.total -= buf.numel()
            self.total += size
        return torch.empty(size, dtype=torch.uint8).pin_memory()

    def give(self, buf, ev):
        with self.lock:
            self.free.append((buf, ev))
            if len(self.free) > 16:      # drop the smallest finished one
                done = [i for i, (b, e) in enumerate(self.free) if e is None or e.query()]
                if done:
                    self.total -= self.free.pop(min(done, key=lambda i: self.free[i][0].numel()))[0].numel()


_PINNED = _PinnedRing()


def _upload(array, dtype, dev):
    """Host array -> device tensor: staged in pinned memory and copied asynchronously on the copy
    stream; the compute stream is ordered after an EVENT recorded right behind this copy — not
    after the whole copy stream, which also carries the input pipeline's next-batch upload and
    its prepare kernels (tools/train_loop.py) — and the host does not wait at all."""
    if dev.type != 'cuda' or dtype not in _NP_OF:
        return torch.tensor(array, dtype=dtype, device=dev)     # (bool, float64, int16, ...: plain path)
    a = np.require(np.asarray(array, dtype=_NP_OF[dtype]), requirements='C')    # keeps a 0-d shape
    side = copy_stream(dev)
    main = torch.cuda.current_stream(dev)
    n = a.nbytes
    with torch.cuda.stream(side):
        t = torch.empty(a.shape, dtype=dtype, device=dev)
        if n > 0:
            stage = _PINNED.take(n)
            stage[:n].numpy()[:] = a.reshape(-1).view(np.uint8)
            t.view(-1).view(torch.uint8).copy_(stage[:n], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(side)
    if n > 0:
        _PINNED.give(stage, ev)
    main.wait_event(ev)
    t.record_stream(main)
    return t


def _pool():
    global _POOL
    if _POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        _POOL = ThreadPoolExecutor(max_workers=4, thread_name_prefix='mrcnn-targets')
    return _POOL


def _upload_many(arrays, dtypes, dev):
    """Several host arrays -> device tensors with ONE copy: the arrays are packed (64-byte
    aligned) into one byte buffer, uploaded through ``_upload`` and returned as typed views."""
    parts, spans, off = [], [], 0
    for a, dt in zip(arrays, dtypes):
        a = np.ascontiguousarray(a, dtype=_NP_OF[dt])
        pad = (-off) % 64
        if pad:
            parts.append(np.zeros(pad, np.uint8))
            off += pad
        parts.append(a.reshape(-1).view(np.uint8))
        spans.append((off, a.nbytes, a.shape, dt))
        off += a.nbytes
    flat = _upload(np.concatenate(parts), torch.uint8, dev)
    return [flat[o:o + n].view(dt).reshape(shape) for o, n, shape, dt in spans]


class MaskRCNNTrainChain(torch.nn.Module):

    def __init__(self, mask_rcnn, rpn_sigma=3., roi_sigma=1.,
                 anchor_target_creator=None, proposal_target_creator=None):
        super(MaskRCNNTrainChain, self).__init__()
        self.mask_rcnn = mask_rcnn
        self.rpn_sigma = rpn_sigma
        self.roi_sigma = roi_sigma
        self.anchor_target_creator = anchor_target_creator or AnchorTargetCreator()
        self.proposal_target_creator = proposal_target_creator or ProposalTargetCreator()
        self.loc_normalize_mean = mask_rcnn.loc_normalize_mean
        self.loc_normalize_std = mask_rcnn.loc_normalize_std
        self.report = {}
        self.features_grad_hook = None     # set by parallel.DataParallelGradSync
        self.mask_branch_fg_only = True
        # run the arithmetic of both target creators as HIP kernels
        # (IoU matrices, label rules, regression targets, mask targets when the ground-truth
        # masks are device tensors); the np.random draws stay on the host, in the reference's
        # order, so the sampled sets are identical.  Off by default: at COCO sizes the host
        # creators already overlap with the GPU and the device path adds read-backs.
        self.device_targets = False
        # The RPN losses ignore all but the sampled anchors, so conv1 of the RPN receives a gradient
        # that is exactly zero outside <= 256 positions per image: its backward runs on those rows
        # only (same sums, 6 % of the dense GEMM work; functions/conv.py SparseRows).  Host-target
        # path only (the positions are known on the host there without a read-back).
        self.sparse_rpn_backward = True
        self.host_timeline = None          # developer aid: list of (label, perf_counter) marks
        # The image batch of the NEXT iteration (device tensor, or a callable returning it / None),
        # when the caller already has it — a resident batch, the input pipeline's prefetched one.
        # Its frozen prefix then runs beside this step's backbone backward.  None: off.
        self.next_imgs = None

    def forward(self, imgs, bboxes, labels, masks, scales):
        """imgs (N,3,H,W) device tensor; bboxes / labels / masks: per-image sequences of
        host (or device) arrays (G,4) f32 / (G,) i32 / (G,H,W) i32; scales (N,) floats."""
        tl = self.host_timeline
        mark = (lambda label: tl.append((label, time.perf_counter()))) if tl is not None \
            else (lambda label: None)
        mark('begin')
        scales = np.asarray([float(s) for s in scales], dtype=np.float32)
        to_np = lambda a: a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
        bboxes = [to_np(b).astype(np.float32) for b in bboxes]
        labels = [to_np(l).astype(np.int32) for l in labels]
        dev = imgs.device

        batch_size, _, H, W = imgs.shape
        img_size = (H, W)

        features = self.mask_rcnn.extractor(imgs)
        if self.features_grad_hook is not None and features.requires_grad:
            # fires once the head's and the RPN's backward are both done
            features.register_hook(self.features_grad_hook)
        nxt = self.next_imgs
        if nxt is not None and features.requires_grad and hasattr(self.mask_rcnn.extractor, 'prefetch_frozen'):
            # the NEXT batch's frozen prefix (conv1 .. res2) is queued on a side stream at the
            # moment the backbone's backward begins (models/resnet_extractor.py:prefetch_frozen)
            def _prefetch(g, nxt=nxt, ext=self.mask_rcnn.extractor):
                t = nxt() if callable(nxt) else nxt
                if t is not None:
                    ext.prefetch_frozen(t)
                return None
            features.register_hook(_prefetch)
        # The deterministic half of the RPN target assignment needs only the ground truth:
        # start it on worker threads now (NumPy releases the GIL) so it overlaps with the
        # GPU's extractor/RPN/head work; its np.random draws happen later, in order.
        anchor_h = self.mask_rcnn.rpn.host_anchor(features.shape[2], features.shape[3], dev)
        atc = self.anchor_target_creator
        ptc = self.proposal_target_creator
        if self.device_targets and dev.type == 'cuda' and hasattr(ptc, 'sample_device') \
                and hasattr(atc, 'prepare_device'):
            (rpn_locs, rpn_scores, sample_rois, sample_roi_indices, gt_roi_locs, gt_roi_labels,
             gt_roi_masks, gt_rpn_locs, gt_rpn_labels, roi_cls_locs, roi_scores, roi_masks,
             mask_rows) = self._forward_device_targets(
                features, img_size, scales, bboxes, labels, masks, anchor_h, mark)
        else:
            (rpn_locs, rpn_scores, sample_rois, sample_roi_indices, gt_roi_locs, gt_roi_labels,
             gt_roi_masks, gt_rpn_locs, gt_rpn_labels, roi_cls_locs, roi_scores, roi_masks,
             mask_rows) = self._forward_host_targets(
                features, img_size, scales, bboxes, labels, masks, anchor_h, mark)
        rpn_locs = rpn_locs.reshape(-1, 4)
        rpn_scores = rpn_scores.reshape(-1)
        rpn_loc_loss = F.fast_rcnn_loc_loss(rpn_locs, gt_rpn_locs, gt_rpn_labels, self.rpn_sigma)
        rpn_cls_loss = F.sigmoid_cross_entropy(rpn_scores, gt_rpn_labels)

        # Losses for outputs of the head: the class-specific 4-vector is selected inside
        # the kernel (roi_cls_locs[arange(n), gt_roi_labels], :168-170).
        roi_loc_loss = F.fast_rcnn_loc_loss(
            roi_cls_locs, gt_roi_locs, gt_roi_labels, self.roi_sigma, cls=gt_roi_labels)
        roi_cls_loss = F.softmax_cross_entropy(roi_scores, gt_roi_labels)

        # Losses for outputs of mask branch (:176-178)
        if mask_rows is not None:
            roi_mask_loss = F.mask_sigmoid_cross_entropy(
                roi_masks, gt_roi_labels.index_select(0, mask_rows),
                gt_roi_masks.index_select(0, mask_rows))
        else:
            roi_mask_loss = F.mask_sigmoid_cross_entropy(roi_masks, gt_roi_labels, gt_roi_masks)

        loss = rpn_loc_loss + rpn_cls_loss + roi_loc_loss + roi_cls_loss + roi_mask_loss
        self.report = {'rpn_loc_loss': rpn_loc_loss.detach(),
                       'rpn_cls_loss': rpn_cls_loss.detach(),
                       'roi_loc_loss': roi_loc_loss.detach(),
                       'roi_cls_loss': roi_cls_loss.detach(),
                       'roi_mask_loss': roi_mask_loss.detach(),
                       'loss': loss.detach()}
        mark('losses queued')
        self.last_targets = {'sample_rois': sample_rois, 'sample_roi_indices': sample_roi_indices,
                             'feature_shape': tuple(features.shape), 'gt_roi_labels': gt_roi_labels,
                             'gt_roi_masks': gt_roi_masks, 'gt_rpn_labels': gt_rpn_labels,
                             'n_rois': int(sample_rois.shape[0]),
                             'n_fg': getattr(self, '_last_n_fg', None)}   # host count, no read-back
        return loss

    def _forward_host_targets(self, features, img_size, scales, bboxes, labels, masks, anchor_h, mark):
        """RPN, host target creators (the reference's arrangement, :126-158) and the RoI head."""
        dev = features.device
        batch_size = features.shape[0]
        to_np = lambda a: a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
        atc = self.anchor_target_creator
        atc_jobs = None
        if hasattr(atc, 'prepare') and hasattr(atc, 'finish'):
            atc_jobs = [_pool().submit(atc.prepare, bbox, anchor_h, img_size) for bbox in bboxes]
        pc = getattr(self.mask_rcnn.rpn, 'proposal_layer', None)
        has_flag = pc is not None and hasattr(pc, 'keep_host_copy')
        prev_flag = pc.keep_host_copy if has_flag else None
        try:
            if has_flag:
                pc.keep_host_copy = True   # proposals also as host arrays, same synchronisation
            rpn_locs, rpn_scores, rois, roi_indices, anchor = self.mask_rcnn.rpn(
                features, img_size, scales)
        finally:
            if has_flag:                   # predict() / eval calls keep their single D2H read
                pc.keep_host_copy = prev_flag

        # proposal targets: host-side sampling, exactly as the reference (:126-146)
        mark('extractor+rpn queued')
        host_rois = getattr(pc, 'last_host_rois', None) if pc is not None else None
        if host_rois is not None:
            pc.last_host_rois = None       # consumed
        if host_rois is None:
            rois_h = rois.cpu().numpy()
            roi_indices_h = roi_indices.cpu().numpy()
            host_rois = [rois_h[roi_indices_h == i] for i in range(batch_size)]
        mark('rois on host')
        ptc = self.proposal_target_creator
        split = hasattr(ptc, 'sample') and hasattr(ptc, 'mask_targets')
        sample_rois, sample_roi_indices = [], []
        gt_roi_locs, gt_roi_labels, gt_roi_masks, mask_jobs = [], [], [], []
        for batch_index, (bbox, label, mask) in enumerate(zip(bboxes, labels, masks)):
            roi = host_rois[batch_index]
            if split:
                sample_roi, gt_roi_loc, gt_roi_label, job = ptc.sample(roi, bbox, label)
                mask_jobs.append((job, mask))
            else:
                sample_roi, gt_roi_loc, gt_roi_label, gt_roi_mask = \
                    ptc(roi, bbox, label, to_np(mask))
                gt_roi_masks.append(gt_roi_mask)
            mark('ptc.sample')
            sample_rois.append(sample_roi)
            sample_roi_indices.append(np.full((len(sample_roi),), batch_index, dtype=np.int32))
            gt_roi_locs.append(gt_roi_loc)
            gt_roi_labels.append(gt_roi_label)
        up = lambda parts, dt: _upload(np.concatenate(parts, axis=0), dt, dev)
        gt_roi_labels_h = np.concatenate(gt_roi_labels, axis=0)
        self._last_n_fg = int((gt_roi_labels_h > 0).sum())
        fg_rows = np.flatnonzero(gt_roi_labels_h > 0) if self.mask_branch_fg_only else np.zeros(0)
        cat = lambda parts: np.concatenate(parts, axis=0)
        # row -> position among the foreground rows (-1: background), for the fused res5 tail
        fg_slot = np.full((len(gt_roi_labels_h),), -1, np.int32)
        fg_slot[fg_rows.astype(np.int64)] = np.arange(len(fg_rows), dtype=np.int32)
        # processing order of the RoIs for the ROIAlign forward (same values in any order)
        from ..functions.roi_align_2d import spatial_order
        roi_order = spatial_order(cat(sample_rois), cat(sample_roi_indices), self.mask_rcnn.head.spatial_scale)
        # the (batch index, x1, y1, x2, y2) rows ROIAlign reads, built here instead of by three small
        # device launches (cat, cast, column permutation) in front of the head
        rois_h, idx_h = cat(sample_rois), cat(sample_roi_indices)
        rois5_h = np.concatenate([idx_h[:, None].astype(np.float32), rois_h[:, [1, 0, 3, 2]]], axis=1)
        (sample_rois, sample_roi_indices, gt_roi_locs, gt_roi_labels, fg_rows_d, fg_slot_d,
         roi_order_d, rois5_d) = _upload_many(
            [rois_h, idx_h, cat(gt_roi_locs), gt_roi_labels_h, fg_rows, fg_slot,
             roi_order, rois5_h],
            [torch.float32, torch.int32, torch.float32, torch.int32, torch.int64, torch.int32,
             torch.int32, torch.float32], dev)
        fg_rows_d._mrcnn_slot = fg_slot_d
        sample_rois._mrcnn_order = roi_order_d
        sample_rois._mrcnn_rois5 = rois5_d

        # The reference runs the mask branch on every sampled RoI (:147-148) although
        # background rows carry all-ignored (-1) mask targets and therefore contribute
        # neither to the loss nor to any gradient.  With
        # ``mask_branch_fg_only`` the branch runs on the foreground rows only: identical loss
        # (same normaliser: the count of non-ignored target pixels) and identical gradients.
        mark('rois sampled')
        mask_rows = None
        if self.mask_branch_fg_only and len(fg_rows) > 0:
            mask_rows = fg_rows_d
        roi_cls_locs, roi_scores, roi_masks = self.mask_rcnn.head(
            features, sample_rois, sample_roi_indices, mask_rows=mask_rows)

        mark('head queued')
        # the head is now queued on the GPU: build the 14x14 mask targets on the host meanwhile
        if split:
            gt_roi_masks = [ptc.mask_targets(job, to_np(mask)) for job, mask in mask_jobs]
        gt_roi_masks = up(gt_roi_masks, torch.int32)
        mark('mask targets')

        # RPN targets (host) — after all ProposalTargetCreator calls, as in the reference,
        # so the global np.random stream is consumed in the same order (:150-158).
        gt_rpn_locs, gt_rpn_labels = [], []
        for i, bbox in enumerate(bboxes):
            if atc_jobs is not None:
                gt_rpn_loc, gt_rpn_label = atc.finish(atc_jobs[i].result())
            else:
                gt_rpn_loc, gt_rpn_label = atc(bbox, anchor_h, img_size)
            gt_rpn_locs.append(gt_rpn_loc)
            gt_rpn_labels.append(gt_rpn_label)
        # Both RPN losses ignore every anchor with label -1 (:150-166), so the gradient of the RPN's
        # conv1 output is exactly zero outside the map positions of the sampled anchors: hand
        # those positions to its backward (functions/conv.py: SparseRows; anchor index =
        # position * A + a, utils/bbox.py enumerate_shifted_anchor)
        extra, extra_dt = [], []
        rpn = self.mask_rcnn.rpn
        hint = getattr(rpn, 'grad_rows', None)
        n_rows = 0
        if hint is not None and self.sparse_rpn_backward:
            from ..functions.conv import SparseRows
            hw = int(features.shape[2] * features.shape[3])
            pos = [np.flatnonzero(np.asarray(lab) >= 0) // rpn.n_anchor + i * hw
                   for i, lab in enumerate(gt_rpn_labels)]
            rows_h, lookup_h = SparseRows.host_tables(np.concatenate(pos), hw * len(gt_rpn_labels))
            n_rows = len(rows_h)
            extra, extra_dt = [rows_h, lookup_h], [torch.int32, torch.int32]
        up_all = _upload_many(
            [np.concatenate(gt_rpn_locs, axis=0), np.concatenate(gt_rpn_labels, axis=0)] + extra,
            [torch.float32, torch.int32] + extra_dt, dev)
        gt_rpn_locs, gt_rpn_labels = up_all[0], up_all[1]
        if n_rows > 0:
            hint.set(up_all[2], up_all[3], n_rows)
        mark('rpn targets')
        return (rpn_locs, rpn_scores, sample_rois, sample_roi_indices, gt_roi_locs, gt_roi_labels,
                gt_roi_masks, gt_rpn_locs, gt_rpn_labels, roi_cls_locs, roi_scores, roi_masks, mask_rows)

    def _forward_device_targets(self, features, img_size, scales, bboxes, labels, masks, anchor_h,
                                mark):
        """Same step with the target arithmetic on the device.  The
        np.random call order is the reference's: every ProposalTargetCreator draw (foreground,
        background, image by image), then every AnchorTargetCreator draw."""
        dev = features.device
        batch_size = features.shape[0]
        atc, ptc = self.anchor_target_creator, self.proposal_target_creator
        anchor_d = self.mask_rcnn.rpn._anchor(features.shape[2], features.shape[3], dev)[1]
        # RPN label rule: needs only the ground truth -> queued ahead of the RPN itself; the
        # labels travel to pinned host memory asynchronously
        atc_states = [atc.prepare_device(bbox, anchor_d, anchor_h, img_size, upload=_upload)
                      for bbox in bboxes]
        rpn_locs, rpn_scores, rois, roi_indices, anchor = self.mask_rcnn.rpn(
            features, img_size, scales)
        mark('extractor+rpn queued')
        counts = getattr(self.mask_rcnn.rpn, 'last_counts', None)
        if counts is None:
            counts = np.bincount(roi_indices.cpu().numpy(), minlength=batch_size).tolist()
        bounds = np.concatenate([[0], np.cumsum(counts)]).astype(int)
        mark('rois on host')
        s_rois, s_idx, g_locs, g_labels, jobs = [], [], [], [], []
        for i, (bbox, label) in enumerate(zip(bboxes, labels)):
            roi_i = rois[int(bounds[i]):int(bounds[i + 1])]
            sample_roi, gt_roi_loc, gt_roi_label, job = ptc.sample_device(roi_i, bbox, label,
                                                                          upload=_upload)
            mark('ptc.sample')
            s_rois.append(sample_roi)
            s_idx.append(torch.full((job['n'],), i, dtype=torch.int32, device=dev))
            g_locs.append(gt_roi_loc)
            g_labels.append(gt_roi_label)
            jobs.append(job)
        sample_rois, sample_roi_indices = torch.cat(s_rois, 0), torch.cat(s_idx, 0)
        gt_roi_locs, gt_roi_labels = torch.cat(g_locs, 0), torch.cat(g_labels, 0)
        offs = np.concatenate([[0], np.cumsum([j['n'] for j in jobs])]).astype(int)
        self._last_n_fg = int(sum(j['n_fg'] for j in jobs))
        fg_rows = np.concatenate([np.arange(offs[i], offs[i] + j['n_fg']) for i, j in enumerate(jobs)]) \
            if self.mask_branch_fg_only else np.zeros(0, np.int64)
        # ground-truth masks that live on the host: their 14x14 targets are built there (as in
        # the host path) from the foreground boxes, read back BEFORE the head is queued
        host_mask = [not (isinstance(m, torch.Tensor) and m.is_cuda) for m in masks]
        fg_boxes_h = {}
        for i, j in enumerate(jobs):
            if host_mask[i] and j['n_fg'] > 0:
                fg_boxes_h[i] = (j['sample_roi'][:j['n_fg']].cpu().numpy(),
                                 j['gt_index'][:j['n_fg']].cpu().numpy())
        mark('rois sampled')
        mask_rows = None
        if self.mask_branch_fg_only and len(fg_rows) > 0:
            mask_rows = _upload(np.asarray(fg_rows, np.int64), torch.int64, dev)
        roi_cls_locs, roi_scores, roi_masks = self.mask_rcnn.head(
            features, sample_rois, sample_roi_indices, mask_rows=mask_rows)
        mark('head queued')
        from .utils.proposal_target_creator import _mask_targets
        parts = []
        for i, (j, m) in enumerate(zip(jobs, masks)):
            if not host_mask[i]:
                parts.append(ptc.mask_targets_device(j, m))
                continue
            t = -np.ones((j['n'], ptc.mask_size, ptc.mask_size), dtype=np.int32)
            if j['n_fg'] > 0:
                boxes, gt_index = fg_boxes_h[i]
                t[:j['n_fg']] = _mask_targets(np.round(boxes).astype(np.int32), gt_index,
                                              np.asarray(m), ptc.mask_size)
            parts.append(_upload(t, torch.int32, dev))
        gt_roi_masks = torch.cat(parts, 0)
        mark('mask targets')
        g_rl, g_rlab = zip(*[atc.finish_device(st) for st in atc_states])
        gt_rpn_locs, gt_rpn_labels = torch.cat(g_rl, 0), torch.cat(g_rlab, 0)
        mark('rpn targets')
        return (rpn_locs, rpn_scores, sample_rois, sample_roi_indices, gt_roi_locs, gt_roi_labels,
                gt_roi_masks, gt_rpn_locs, gt_rpn_labels, roi_cls_locs, roi_scores, roi_masks, mask_rows)
