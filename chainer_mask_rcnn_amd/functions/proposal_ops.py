"""Device-side box ops behind the proposal / detection path.

``non_maximum_suppression`` mirrors chainercv's function of the same name (the
reference imports it at /root/reference/chainer_mask_rcnn/models/mask_rcnn.py:39
and uses it at :193-194; ProposalCreator calls it internally).  The other
functions are thin wrappers over the C ABI used by ``ProposalCreator``.
"""
import torch

from .. import _lib


def decode_clip(anchor, loc, img_size, min_size=0., out=None):
    """loc2bbox + clip (+ min-size validity). anchor, loc: (n,4) -> roi (n,4), valid (n,) uint8
    (``out`` = preallocated contiguous (roi, valid) to write into)."""
    _lib.require_device(anchor, loc)
    n = anchor.shape[0]
    anchor = anchor.contiguous()
    loc = loc.contiguous()
    if out is not None:
        roi, valid = out
    else:
        roi = torch.empty((n, 4), dtype=torch.float32, device=loc.device)
        valid = torch.empty((n,), dtype=torch.uint8, device=loc.device)
    _lib.call('mrcnn_decode_clip', _lib.ptr(anchor), _lib.ptr(loc), _lib.ptr(roi),
              _lib.ptr(valid), n, float(img_size[0]), float(img_size[1]),
              float(min_size), _lib.stream_ptr())
    return roi, valid


def topk_desc_batched(score, k, valid=None):
    """Stable descending top-k of every row of score (G, n) (valid (G, n) uint8 or None).
    Returns (order int32 (G, k), n_out int32 (G,))."""
    _lib.require_device(score)
    score = score.contiguous()
    G, n = score.shape
    k = int(min(k, n)) if k > 0 else n
    order = torch.empty((G, max(k, 1)), dtype=torch.int32, device=score.device)
    n_out = torch.empty((G,), dtype=torch.int32, device=score.device)
    per = (_lib.load().mrcnn_topk_workspace_bytes(n) + 63) // 64 * 64
    ws = _lib.workspace(per * G, score.device, 'topk')
    if valid is not None:
        valid = valid.contiguous()
    _lib.call('mrcnn_topk_desc_batched', _lib.ptr(score), _lib.ptr(valid), G, n, k,
              _lib.ptr(order), _lib.ptr(n_out), _lib.ptr(ws), _lib.stream_ptr())
    return order, n_out


def topk_desc(score, k, valid=None):
    """Stable descending top-k. Returns (order int32 (k,), n_out int32 device scalar)."""
    _lib.require_device(score)
    score = score.contiguous()
    n = score.numel()
    k = int(min(k, n)) if k > 0 else n
    order = torch.empty((max(k, 1),), dtype=torch.int32, device=score.device)
    n_out = torch.empty((1,), dtype=torch.int32, device=score.device)
    ws = _lib.workspace(_lib.load().mrcnn_topk_workspace_bytes(n), score.device, 'topk')
    _lib.call('mrcnn_topk_desc', _lib.ptr(score), _lib.ptr(valid), n, k,
              _lib.ptr(order), _lib.ptr(n_out), _lib.ptr(ws), _lib.stream_ptr())
    return order[:k], n_out


def gather_rows(src, idx, n_dev=None):
    """dst[j] = src[idx[j]] for j < n_dev (zero rows after)."""
    src = src.contiguous()
    cols = src.shape[1]
    n_max = idx.numel()
    dst = torch.empty((n_max, cols), dtype=torch.float32, device=src.device)
    _lib.call('mrcnn_gather_rows', _lib.ptr(src), _lib.ptr(idx), _lib.ptr(n_dev),
              n_max, cols, _lib.ptr(dst), _lib.stream_ptr())
    return dst


def nms_sorted(bbox, thresh, n_dev=None, limit=0):
    """NMS over score-sorted boxes. Returns (keep int32 (n_max,), n_keep int32 device scalar)."""
    _lib.require_device(bbox)
    bbox = bbox.contiguous()
    n_max = bbox.shape[0]
    keep = torch.empty((max(n_max, 1),), dtype=torch.int32, device=bbox.device)
    n_keep = torch.empty((1,), dtype=torch.int32, device=bbox.device)
    ws = _lib.workspace(_lib.load().mrcnn_nms_workspace_bytes(n_max, 1), bbox.device, 'nms')
    _lib.call('mrcnn_nms_sorted', _lib.ptr(bbox), _lib.ptr(n_dev), n_max, float(thresh),
              int(limit or 0), _lib.ptr(keep), _lib.ptr(n_keep), _lib.ptr(ws),
              _lib.stream_ptr())
    return keep, n_keep


def nms_sorted_batched(bbox, n_dev, thresh, limit=0):
    """bbox (G, n_max, 4) sorted per group, n_dev (G,) int32. Returns keep (G,n_max), n_keep (G,)."""
    _lib.require_device(bbox, n_dev)
    bbox = bbox.contiguous()
    G, n_max = bbox.shape[0], bbox.shape[1]
    keep = torch.empty((G, max(n_max, 1)), dtype=torch.int32, device=bbox.device)
    n_keep = torch.empty((G,), dtype=torch.int32, device=bbox.device)
    ws = _lib.workspace(_lib.load().mrcnn_nms_workspace_bytes(n_max, G), bbox.device, 'nms')
    _lib.call('mrcnn_nms_sorted_batched', _lib.ptr(bbox), _lib.ptr(n_dev), G, n_max,
              float(thresh), int(limit or 0), _lib.ptr(keep), _lib.ptr(n_keep),
              _lib.ptr(ws), _lib.stream_ptr())
    return keep, n_keep


def non_maximum_suppression(bbox, thresh, score=None, limit=None):
    """Suppress bounding boxes according to their IoUs (chainercv semantics).

    bbox: (R, 4) float32 device tensor (y_min, x_min, y_max, x_max).  Returns an
    int32 device tensor of selected indices, sorted by descending score when
    ``score`` is given (ties: lower index first).
    """
    _lib.require_device(bbox)
    if bbox.shape[0] == 0:
        return torch.zeros((0,), dtype=torch.int32, device=bbox.device)
    order = None
    if score is not None:
        order, _ = topk_desc(score, bbox.shape[0])
        bbox = gather_rows(bbox, order)
    keep, n_keep = nms_sorted(bbox, thresh, None, limit or 0)
    keep = keep[:int(n_keep.item())]
    if order is not None:
        keep = order[keep.long()]
    return keep
