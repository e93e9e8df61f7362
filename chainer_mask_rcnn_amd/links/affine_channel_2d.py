"""AffineChannel2D link — drop-in for
/root/reference/chainer_mask_rcnn/links/affine_channel_2d.py:8-23."""
import torch

from ..functions import affine_channel_2d


class AffineChannel2D(torch.nn.Module):

    """A simple channel-wise affine transformation operation.

    Parameters ``W`` (ones) and ``b`` (zeros) of shape ``(channels,)``.  Inside the
    model these are consumed as the fused scale/shift of the preceding
    convolution's epilogue; calling the link applies the stand-alone HIP op.
    """

    def __init__(self, channels):
        super(AffineChannel2D, self).__init__()
        self.W = torch.nn.Parameter(torch.ones(channels, dtype=torch.float32))
        self.b = torch.nn.Parameter(torch.zeros(channels, dtype=torch.float32))

    def forward(self, x):
        W = self.W.reshape(1, -1, 1, 1)
        b = self.b.reshape(1, -1, 1, 1)
        return affine_channel_2d(x, W, b)
