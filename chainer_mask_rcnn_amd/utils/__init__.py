# flake8: noqa
from .bbox import (generate_anchor_base, enumerate_shifted_anchor, bbox_iou, bbox2loc,
                   resize_bilinear)
