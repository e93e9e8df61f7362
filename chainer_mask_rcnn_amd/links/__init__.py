"""Parameterised layers of the hot path.  The only link the reference defines itself is the
frozen-BatchNorm replacement; convolutions live in ``models.resnet_extractor``."""
from . import affine_channel_2d as _acl

AffineChannel2D = _acl.AffineChannel2D

__all__ = ['AffineChannel2D']
