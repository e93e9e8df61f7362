"""AffineChannel2D — drop-in for the reference's
``chainer_mask_rcnn.functions.affine_channel_2d``
(/root/reference/chainer_mask_rcnn/functions/affine_channel_2d.py:8-66).

Inside the model the affine is fused into the convolution epilogue
(functions/conv.py); this stand-alone function keeps the public API and is what
the reference's own unit test exercises.
"""
import torch

from .. import _lib
from ._layout import nhwc, empty_nhwc


class AffineChannel2DFunction(torch.autograd.Function):

    @staticmethod
    def forward(ctx, x, W, b):
        _lib.require_device(x, W, b)
        # check_type_forward, affine_channel_2d.py:24-36
        if not (x.is_floating_point() and W.is_floating_point() and b.is_floating_point()
                and x.dim() == 4 and W.dim() == 4 and b.dim() == 4
                and W.shape[1] == b.shape[1]):
            raise TypeError('affine_channel_2d expects x (N,C,H,W), W and b (1,C,1,1) floats')
        x = nhwc(x)
        N, C, H, Wd = x.shape
        Wf = W.reshape(-1).contiguous()
        bf = b.reshape(-1).contiguous()
        y = empty_nhwc((N, C, H, Wd), x.device)
        _lib.call('mrcnn_affine_fwd', _lib.ptr(x), _lib.ptr(Wf), _lib.ptr(bf), _lib.ptr(y),
                  N * H * Wd, C, _lib.stream_ptr())
        ctx.save_for_backward(x, Wf)
        ctx.wshape = tuple(W.shape)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, Wf = ctx.saved_tensors
        gy = nhwc(gy)
        N, C, H, Wd = x.shape
        gx = empty_nhwc((N, C, H, Wd), gy.device)
        gW = torch.empty((C,), dtype=torch.float32, device=gy.device)
        gb = torch.empty((C,), dtype=torch.float32, device=gy.device)
        ws = _lib.workspace(_lib.load().mrcnn_colsum_workspace_bytes(C), gy.device, 'colsum')
        _lib.call('mrcnn_affine_bwd', _lib.ptr(x), _lib.ptr(Wf), _lib.ptr(gy), _lib.ptr(gx),
                  _lib.ptr(gW), _lib.ptr(gb), N * H * Wd, C, _lib.ptr(ws), _lib.stream_ptr())
        return gx, gW.reshape(ctx.wshape), gb.reshape(ctx.wshape)


def affine_channel_2d(x, W, b):
    return AffineChannel2DFunction.apply(x, W, b)
