"""The Mask R-CNN losses on HIP kernels (no host synchronisation).

Mirrors chainer's ``F.sigmoid_cross_entropy`` / ``F.softmax_cross_entropy`` and the
reference's ``_fast_rcnn_loc_loss`` / ``_smooth_l1_loss``
(/root/reference/chainer_mask_rcnn/models/mask_rcnn_train_chain.py:163-181,
:192-213; formulas SURVEY.md Appendix A.1).  Each returns a 0-dim device tensor.
"""
import torch

from .. import _lib


def _ws(device, rows=0):
    return _lib.workspace(_lib.load().mrcnn_loss_workspace_bytes(rows), device, 'loss')


def _scalar(device):
    return torch.empty((), dtype=torch.float32, device=device)


class _SigmoidCEFn(torch.autograd.Function):

    @staticmethod
    def forward(ctx, x, t):
        _lib.require_device(x, t)
        xc = x.contiguous()
        t = t.contiguous()
        loss = _scalar(x.device)
        gx = torch.empty_like(xc) if x.requires_grad else None
        _lib.call('mrcnn_sigmoid_ce', _lib.ptr(xc), _lib.ptr(t), xc.numel(), _lib.ptr(loss),
                  _lib.ptr(gx), _lib.ptr(_ws(x.device)), _lib.stream_ptr())
        ctx.gx = gx
        ctx.xshape = x.shape
        return loss

    @staticmethod
    def backward(ctx, g):
        return ctx.gx.reshape(ctx.xshape) * g, None


def sigmoid_cross_entropy(x, t):
    """F.sigmoid_cross_entropy(x, t): t int32 in {-1,0,1}, -1 ignored, mean over the rest."""
    if t.dtype != torch.int32:
        raise TypeError('sigmoid_cross_entropy: t must be int32')
    if x.shape != t.shape:
        raise ValueError('sigmoid_cross_entropy: shape mismatch %s vs %s' % (x.shape, t.shape))
    return _SigmoidCEFn.apply(x, t)


class _MaskSigmoidCEFn(torch.autograd.Function):

    @staticmethod
    def forward(ctx, roi_masks, gt_label, gt_mask):
        from ._layout import nhwc, empty_nhwc
        _lib.require_device(roi_masks, gt_label, gt_mask)
        x = nhwc(roi_masks)
        R, Kc, H, W = x.shape
        loss = _scalar(x.device)
        gx = empty_nhwc((R, Kc, H, W), x.device) if roi_masks.requires_grad else None
        # bind temporaries to names: a tensor created inside the argument list would be freed
        # (and its block possibly re-used) before the asynchronous kernel reads it
        gt_label, gt_mask = gt_label.contiguous(), gt_mask.contiguous()
        _lib.call('mrcnn_mask_sigmoid_ce', _lib.ptr(x), _lib.ptr(gt_label),
                  _lib.ptr(gt_mask), R, H * W, Kc, _lib.ptr(loss), _lib.ptr(gx),
                  _lib.ptr(_ws(x.device)), _lib.stream_ptr())
        ctx.gx = gx
        return loss

    @staticmethod
    def backward(ctx, g):
        return ctx.gx * g, None, None


def mask_sigmoid_cross_entropy(roi_masks, gt_roi_labels, gt_roi_masks):
    """``F.sigmoid_cross_entropy(roi_masks[arange(n), gt_roi_labels - 1], gt_roi_masks)``
    (models/mask_rcnn_train_chain.py:176-178) without materialising the gather."""
    return _MaskSigmoidCEFn.apply(roi_masks, gt_roi_labels, gt_roi_masks)


class _SoftmaxCEFn(torch.autograd.Function):

    @staticmethod
    def forward(ctx, x, t):
        _lib.require_device(x, t)
        assert x.dim() == 2 and x.stride(1) == 1
        R, ncls = x.shape
        loss = _scalar(x.device)
        gx = torch.empty((R, ncls), dtype=torch.float32, device=x.device) \
            if x.requires_grad else None
        t = t.contiguous()
        _lib.call('mrcnn_softmax_ce', _lib.ptr(x), x.stride(0), _lib.ptr(t), R,
                  ncls, _lib.ptr(loss), _lib.ptr(gx), ncls, _lib.ptr(_ws(x.device, R)),
                  _lib.stream_ptr())
        ctx.gx = gx
        return loss

    @staticmethod
    def backward(ctx, g):
        return ctx.gx * g, None


def softmax_cross_entropy(x, t):
    """F.softmax_cross_entropy(x (R,ncls), t int32), ignore_label=-1."""
    if t.dtype != torch.int32:
        raise TypeError('softmax_cross_entropy: t must be int32')
    if x.stride(1) != 1:
        x = x.contiguous()
    return _SoftmaxCEFn.apply(x, t)


class _SmoothL1Fn(torch.autograd.Function):

    @staticmethod
    def forward(ctx, pred, gt_loc, gt_label, sigma, cls):
        _lib.require_device(pred, gt_loc, gt_label)
        assert pred.dim() == 2
        pc = pred.contiguous()
        gt_loc, gt_label = gt_loc.contiguous(), gt_label.contiguous()
        n, width = pc.shape
        loss = _scalar(pred.device)
        # the kernel writes only the selected 4-vectors of rows with label > 0
        gx = torch.zeros((n, width), dtype=torch.float32, device=pred.device) \
            if pred.requires_grad else None
        _lib.call('mrcnn_smooth_l1', _lib.ptr(pc), width, _lib.ptr(cls),
                  _lib.ptr(gt_loc), _lib.ptr(gt_label), n,
                  float(sigma), _lib.ptr(loss), _lib.ptr(gx), _lib.ptr(_ws(pred.device)),
                  _lib.stream_ptr())
        ctx.gx = gx
        return loss

    @staticmethod
    def backward(ctx, g):
        return ctx.gx * g, None, None, None, None


def fast_rcnn_loc_loss(pred_loc, gt_loc, gt_label, sigma, cls=None):
    """``_fast_rcnn_loc_loss`` (models/mask_rcnn_train_chain.py:204-213).

    pred_loc (n,4), or (n, 4*n_class) together with ``cls`` (n,) int32 selecting the
    4-vector of class ``cls[i]`` per row (:169-170) without materialising the gather.
    """
    if gt_label.dtype != torch.int32:
        raise TypeError('fast_rcnn_loc_loss: gt_label must be int32')
    return _SmoothL1Fn.apply(pred_loc, gt_loc, gt_label, sigma, cls)


def softmax(x):
    """F.softmax over axis 1 of (R, ncls) (models/mask_rcnn.py:208)."""
    _lib.require_device(x)
    if x.stride(1) != 1:
        x = x.contiguous()
    R, ncls = x.shape
    y = torch.empty((R, ncls), dtype=torch.float32, device=x.device)
    _lib.call('mrcnn_softmax', _lib.ptr(x), x.stride(0), _lib.ptr(y), ncls, R, ncls,
              _lib.stream_ptr())
    return y
