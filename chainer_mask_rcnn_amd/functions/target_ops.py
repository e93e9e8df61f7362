"""Thin wrappers over the device half of the target creators (include/mrcnn_hip.h "Target
creators: the device half")."""
import ctypes

import torch

from .. import _lib


def bbox_iou_argmax(boxes_a, boxes_b, want_matrix=False):
    """chainercv ``bbox_iou(a, b)`` reduced per row of ``a``: (max_iou (na,) f32, argmax (na,)
    i32[, iou (na,g), col_max (g,)])."""
    _lib.require_device(boxes_a, boxes_b)
    boxes_a, boxes_b = boxes_a.contiguous(), boxes_b.contiguous()
    na, g = boxes_a.shape[0], boxes_b.shape[0]
    dev = boxes_a.device
    best = torch.empty((na,), dtype=torch.float32, device=dev)
    arg = torch.empty((na,), dtype=torch.int32, device=dev)
    iou = col = None
    if want_matrix:
        iou = torch.empty((na, g), dtype=torch.float32, device=dev)
        col = torch.empty((g,), dtype=torch.float32, device=dev)
    _lib.call('mrcnn_bbox_iou_argmax', _lib.ptr(boxes_a), na, _lib.ptr(boxes_b), g, _lib.ptr(iou),
              _lib.ptr(best), _lib.ptr(arg), _lib.ptr(col), _lib.stream_ptr())
    return (best, arg, iou, col) if want_matrix else (best, arg)


def anchor_labels(iou, max_iou, gt_max, neg_iou_thresh, pos_iou_thresh):
    na, g = iou.shape
    label = torch.empty((na,), dtype=torch.int32, device=iou.device)
    _lib.call('mrcnn_anchor_labels', _lib.ptr(iou), _lib.ptr(max_iou), _lib.ptr(gt_max), na, g,
              float(neg_iou_thresh), float(pos_iou_thresh), _lib.ptr(label), _lib.stream_ptr())
    return label


def anchor_targets_finish(anchor_inside, inside_index, label_inside, argmax, bbox, disabled, n_anchor):
    dev = anchor_inside.device
    loc = torch.empty((n_anchor, 4), dtype=torch.float32, device=dev)
    label = torch.empty((n_anchor,), dtype=torch.int32, device=dev)
    n_dis = 0 if disabled is None else int(disabled.numel())
    _lib.call('mrcnn_anchor_targets_finish', _lib.ptr(anchor_inside), _lib.ptr(inside_index),
              _lib.ptr(label_inside), _lib.ptr(argmax), _lib.ptr(bbox), int(anchor_inside.shape[0]),
              _lib.ptr(disabled) if n_dis else None, n_dis, int(n_anchor), _lib.ptr(loc),
              _lib.ptr(label), _lib.stream_ptr())
    return loc, label


def proposal_targets_gather(cand, bbox, gt_label, assigned, chosen, n_fg, mean, std):
    dev = cand.device
    n = int(chosen.numel())
    sample_roi = torch.empty((n, 4), dtype=torch.float32, device=dev)
    loc = torch.empty((n, 4), dtype=torch.float32, device=dev)
    label = torch.empty((n,), dtype=torch.int32, device=dev)
    gt_index = torch.empty((n,), dtype=torch.int32, device=dev)
    m = (_lib.c_f32 * 4)(*[float(v) for v in mean])
    s = (_lib.c_f32 * 4)(*[float(v) for v in std])
    _lib.call('mrcnn_proposal_targets_gather', _lib.ptr(cand), _lib.ptr(bbox), _lib.ptr(gt_label),
              _lib.ptr(assigned), _lib.ptr(chosen), n, int(n_fg), m, s, _lib.ptr(sample_roi),
              _lib.ptr(loc), _lib.ptr(label), _lib.ptr(gt_index), _lib.stream_ptr())
    return sample_roi, loc, label, gt_index


def mask_targets(masks_u8, sample_roi, gt_index, n_fg, mask_size):
    """masks_u8 (G,H,W) uint8 device -> (n, M, M) int32 {-1,0,1}."""
    _lib.require_device(masks_u8, sample_roi)
    assert masks_u8.dtype == torch.uint8 and masks_u8.is_contiguous()
    G, H, W = masks_u8.shape
    n = int(sample_roi.shape[0])
    out = torch.empty((n, mask_size, mask_size), dtype=torch.int32, device=sample_roi.device)
    _lib.call('mrcnn_mask_targets', _lib.ptr(masks_u8), G, H, W, _lib.ptr(sample_roi),
              _lib.ptr(gt_index), n, int(n_fg), int(mask_size), _lib.ptr(out), _lib.stream_ptr())
    return out
