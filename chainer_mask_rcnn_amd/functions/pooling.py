"""Pooling ops of the ResNet-C4 path (NHWC HIP kernels).

``max_pooling_2d`` is chainer's ``F.max_pooling_2d(x, 3, stride=2, pad=1)`` with its
default ``cover_all=True`` (/root/reference/chainer_mask_rcnn/models/resnet_extractor.py:69;
SURVEY.md A.1: 400x667 -> 201x334).  ``average_pooling_2d`` is
``F.average_pooling_2d(res5, 7, stride=7)`` on a 7x7 map (models/mask_rcnn_resnet.py:188).
"""
import torch

from .. import _lib
from ._layout import nhwc, empty_nhwc


def cover_all_out_size(size, k=3, s=2, p=1):
    return (size + 2 * p - k + s - 1) // s + 1


class _MaxPoolFn(torch.autograd.Function):

    @staticmethod
    def forward(ctx, x):
        _lib.require_device(x)
        x = nhwc(x)
        N, C, H, W = x.shape
        P, Q = cover_all_out_size(H), cover_all_out_size(W)
        y = empty_nhwc((N, C, P, Q), x.device)
        _lib.call('mrcnn_maxpool3x3s2p1_fwd', _lib.ptr(x), _lib.ptr(y), N, H, W, C, P, Q,
                  _lib.stream_ptr())
        return y

    @staticmethod
    def backward(ctx, gy):
        raise _lib.MrcnnHipError(
            'pool1 sits below the frozen res2 stage (models/resnet_extractor.py:86-87): '
            'no backward is implemented')


def max_pooling_2d(x, ksize=3, stride=2, pad=1):
    if (ksize, stride, pad) != (3, 2, 1):
        raise ValueError('only the ResNet stem pool (3, stride 2, pad 1, cover_all) is implemented')
    return _MaxPoolFn.apply(x)


class _AvgPoolFn(torch.autograd.Function):

    @staticmethod
    def forward(ctx, x):
        _lib.require_device(x)
        x = nhwc(x)
        R, C, H, W = x.shape
        y = torch.empty((R, C), dtype=torch.float32, device=x.device)
        _lib.call('mrcnn_avgpool_fwd', _lib.ptr(x), _lib.ptr(y), R, H * W, C, _lib.stream_ptr())
        ctx.shape = (R, C, H, W)
        return y.reshape(R, C, 1, 1)

    @staticmethod
    def backward(ctx, gy):
        R, C, H, W = ctx.shape
        gy = gy.reshape(R, C).contiguous()
        gx = empty_nhwc((R, C, H, W), gy.device)
        _lib.call('mrcnn_avgpool_bwd', _lib.ptr(gy), _lib.ptr(gx), R, H * W, C, 0,
                  _lib.stream_ptr())
        return gx


def average_pooling_2d(x, ksize, stride=None):
    """Global average over the (ksize x ksize) map -> (R, C, 1, 1)."""
    if x.shape[2] != ksize or x.shape[3] != ksize:
        raise ValueError('average_pooling_2d is implemented for ksize == map size (7x7 -> 1x1)')
    return _AvgPoolFn.apply(x)
