"""Input pipeline of the train step: the reference's
``datasets/transforms.py`` and ``datasets/concat_examples.py`` with the image half on the
device."""
from .transforms import MaskRCNNTransform  # NOQA
from .transforms import resize_bbox, flip_bbox, resize_nearest, flip  # NOQA
from .concat_examples import concat_examples  # NOQA
