# flake8: noqa
"""Operators — same public names as the reference's ``chainer_mask_rcnn.functions``
(/root/reference/chainer_mask_rcnn/functions/__init__.py:1-9) for the hot path."""
from .affine_channel_2d import affine_channel_2d
from .affine_channel_2d import AffineChannel2DFunction

from .roi_align_2d import roi_align_2d
from .roi_align_2d import ROIAlign2D
from .roi_align_2d import spatial_order as roi_spatial_order

from .conv import conv2d, deconv2x2s2, linear, stem_conv, bottleneck, building_block
from .pooling import max_pooling_2d, average_pooling_2d
from .rows import fanout_rows
from .loss import (sigmoid_cross_entropy, softmax_cross_entropy, fast_rcnn_loc_loss,
                   mask_sigmoid_cross_entropy, softmax)
from .proposal_ops import non_maximum_suppression
