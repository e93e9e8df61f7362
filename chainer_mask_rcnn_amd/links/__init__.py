# flake8: noqa
from .affine_channel_2d import AffineChannel2D
