# flake8: noqa
"""chainer_mask_rcnn_amd — MI355X-native hot path of wkentaro/chainer-mask-rcnn.

Same module layout as the reference package for the path that matters
(``functions``, ``links``, ``models``, ``datasets`` transform/converter); everything numeric runs in hand-written
HIP kernels behind the C ABI of ``include/mrcnn_hip.h``.
"""
__version__ = '0.1.0'

from . import datasets
from . import functions
from . import links
from . import models
from . import utils
