"""Input pipeline of the train step: the reference's
``datasets/coco.py`` (annotation decoding), ``datasets/transforms.py`` and
``datasets/concat_examples.py`` (the image half on the device)."""
from .transforms import MaskRCNNTransform  # NOQA
from .transforms import resize_bbox, flip_bbox, resize_nearest, flip  # NOQA
from .concat_examples import concat_examples  # NOQA
from .coco import COCOInstanceSegmentationDataset  # NOQA
