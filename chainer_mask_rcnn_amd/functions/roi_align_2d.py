"""ROIAlign on MI355X — drop-in for the reference's
``chainer_mask_rcnn.functions.roi_align_2d`` / ``ROIAlign2D``
(/root/reference/chainer_mask_rcnn/functions/roi_align_2d.py:25-60, :527-560).

Same names, argument meaning and error behaviour; tensors are PyTorch-ROCm
tensors with the reference's logical NCHW shapes (physically channels-last),
arithmetic is the hand-written HIP kernel behind ``mrcnn_roi_align_fwd/bwd``.
"""
import torch

from .. import _lib
from ._layout import nhwc, empty_nhwc


# Pixel-owner backward (no atomics, fixed summation order); False = atomic gather form.
DETERMINISTIC_BACKWARD = True
# debug / test switch: check that ``order`` is a permutation of 0..R-1 (one device sort + a read-back)
VALIDATE_ORDER = False


class _ROIAlign2DFn(torch.autograd.Function):

    @staticmethod
    def forward(ctx, x, rois, outh, outw, spatial_scale, sampling_ratio, bin_stride=1, order=None):
        _lib.require_device(x, rois)
        x = nhwc(x)
        rois = rois.contiguous()
        N, C, H, W = x.shape
        R = rois.shape[0]
        oh = (outh + bin_stride - 1) // bin_stride
        ow = (outw + bin_stride - 1) // bin_stride
        y = empty_nhwc((R, C, oh, ow), x.device)
        if order is not None:
            if not (order.dtype == torch.int32 and order.is_contiguous() and order.device == x.device
                    and tuple(order.shape) == (R,)):
                raise TypeError('roi_align_2d: order must be a contiguous int32 device tensor of '
                                'shape (R,) — a permutation of the RoI rows')
            if VALIDATE_ORDER and R > 0 and not torch.equal(
                    torch.sort(order.long())[0], torch.arange(R, device=order.device)):
                raise ValueError('roi_align_2d: order is not a permutation of 0..R-1 (a duplicate '
                                 'leaves output rows unwritten, an out-of-range value reads past rois)')
        _lib.call('mrcnn_roi_align_fwd_ex', _lib.ptr(x), _lib.ptr(rois), _lib.ptr(y),
                  N, H, W, C, R, outh, outw, bin_stride, spatial_scale, sampling_ratio,
                  _lib.ptr(order) if order is not None and R > 0 else None, _lib.stream_ptr())
        # only rois are retained (roi_align_2d.py:62-63 retain_inputs((1,)))
        ctx.save_for_backward(rois)
        ctx.x_shape = (N, C, H, W)
        ctx.args = (outh, outw, spatial_scale, sampling_ratio, bin_stride)
        return y

    @staticmethod
    def backward(ctx, gy):
        rois, = ctx.saved_tensors
        N, C, H, W = ctx.x_shape
        outh, outw, spatial_scale, sampling_ratio, bin_stride = ctx.args
        gy = nhwc(gy)
        gx = empty_nhwc((N, C, H, W), gy.device)
        R = rois.shape[0]
        ws = _lib.workspace(_lib.load().mrcnn_roi_align_bwd_workspace_bytes(
            N, H, W, R, outh, outw, bin_stride), gy.device,
                            'roi_align_bwd') if DETERMINISTIC_BACKWARD else None
        _lib.call('mrcnn_roi_align_bwd_ws', _lib.ptr(gy), _lib.ptr(rois), _lib.ptr(gx),
                  N, H, W, C, R, outh, outw, bin_stride, spatial_scale,
                  sampling_ratio, _lib.ptr(ws),
                  int(ws.numel() * ws.element_size()) if ws is not None else 0, _lib.stream_ptr())
        # no gradient w.r.t. rois (roi_align_2d.py:389, :524)
        return gx, None, None, None, None, None, None, None


class ROIAlign2D(object):

    """ROI align over a set of 2d planes (reference: roi_align_2d.py:25-47)."""

    def __init__(self, outh, outw, spatial_scale, sampling_ratio=0, bin_stride=1, order=None):
        self.bin_stride = int(bin_stride)
        self.order = order
        for arg, value in (('outh', outh), ('outw', outw),
                           ('sampling_ratio', sampling_ratio)):
            if not (isinstance(value, int) and not isinstance(value, bool)
                    and value >= 0):
                raise TypeError(
                    '{} must be positive integer: {}, {}'
                    .format(arg, type(value), value))
        if isinstance(spatial_scale, int):
            spatial_scale = float(spatial_scale)
        elif not isinstance(spatial_scale, float):
            raise TypeError(
                'spatial_scale must be float: {}'.format(type(spatial_scale)))
        self.outh, self.outw = outh, outw
        self.spatial_scale = spatial_scale
        self.sampling_ratio = sampling_ratio

    def check_type_forward(self, x, rois):
        # roi_align_2d.py:49-59
        if not (x.dtype == torch.float32 and x.dim() == 4 and
                rois.dtype == torch.float32 and rois.dim() == 2 and
                rois.shape[1] == 5):
            raise TypeError(
                'ROIAlign2D expects x: float32 (N,C,H,W), rois: float32 (R,5); got '
                '{} {} and {} {}'.format(x.dtype, tuple(x.shape), rois.dtype,
                                         tuple(rois.shape)))

    def __call__(self, x, rois):
        self.check_type_forward(x, rois)
        return _ROIAlign2DFn.apply(x, rois, self.outh, self.outw,
                                   self.spatial_scale, self.sampling_ratio, self.bin_stride,
                                   self.order)


def spatial_order(rois_yx, roi_indices, spatial_scale, band=6.0):
    """Processing order for ``roi_align_2d(..., order=)`` from HOST arrays: RoIs sorted by
    (image, band of ``band`` feature rows of the box centre, x centre).  ``rois_yx`` (R, 4) rows
    ``(y_min, x_min, y_max, x_max)``, ``roi_indices`` (R,).  Returns int32 (R,) NumPy."""
    import numpy as np
    rois_yx = np.asarray(rois_yx, np.float32).reshape(-1, 4)
    yc = (rois_yx[:, 0] + rois_yx[:, 2]) * (0.5 * spatial_scale)
    xc = (rois_yx[:, 1] + rois_yx[:, 3]) * (0.5 * spatial_scale)
    return np.lexsort((xc, np.floor(yc / band), np.asarray(roi_indices))).astype(np.int32)


def roi_align_2d(x, rois, outh, outw, spatial_scale, sampling_ratio=0, axes='xy',
                 bin_stride=1, order=None):
    """Spatial Region of Interest (ROI) align function.

    x: (N, C, H, W) float32; rois: (R, 5) float32 rows
    ``(batch_index, x_min, y_min, x_max, y_max)`` (``axes='xy'``) or
    ``(batch_index, y_min, x_min, y_max, x_max)`` (``axes='yx'``).
    Returns (R, C, outh, outw).  Reference: roi_align_2d.py:527-560.

    ``bin_stride`` (extension, default 1 = reference behaviour): produce only every
    ``bin_stride``-th bin in each direction, i.e. exactly ``roi_align_2d(...)[:, :, ::s, ::s]``.

    ``order`` (extension, default None): int32 device permutation of the RoI rows — the sequence
    in which the forward kernel PROCESSES them (``spatial_order``); the result does not depend on it.
    """
    if axes not in ['xy', 'yx']:
        raise ValueError('Unsupported axes: {}'.format(axes))
    if axes == 'yx':
        rois = rois[:, [0, 2, 1, 4, 3]]
    return ROIAlign2D(outh, outw, spatial_scale, sampling_ratio, bin_stride, order)(x, rois)
