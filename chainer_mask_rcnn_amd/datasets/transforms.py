"""MaskRCNNTransform — /root/reference/chainer_mask_rcnn/datasets/transforms.py:10-51.

Same call protocol and return tuple as the reference.  What moves: the image is uploaded
once as it was decoded (HWC, usually uint8) and resized / mean-subtracted / flipped by ONE
device kernel (``mrcnn_prepare_image``), so the transform returns a device tensor for the
image; boxes, labels and masks stay host NumPy arrays because the target creators that
consume them run on the host (``MaskRCNNTrainChain``).  The random flip consumes one
``random.choice([True, False])`` from Python's global generator, exactly what
``chainercv.transforms.random_flip(img, x_random=True)`` draws.
"""
import random

import numpy as np


def resize_bbox(bbox, in_size, out_size):
    """chainercv.transforms.resize_bbox: scale (y_min, x_min, y_max, x_max) boxes from an
    image of ``in_size`` (H, W) to ``out_size``; float32 in, float32 out."""
    bbox = bbox.copy()
    y_scale = float(out_size[0]) / in_size[0]
    x_scale = float(out_size[1]) / in_size[1]
    bbox[:, 0] = y_scale * bbox[:, 0]
    bbox[:, 2] = y_scale * bbox[:, 2]
    bbox[:, 1] = x_scale * bbox[:, 1]
    bbox[:, 3] = x_scale * bbox[:, 3]
    return bbox


def flip_bbox(bbox, size, y_flip=False, x_flip=False):
    """chainercv.transforms.flip_bbox."""
    H, W = size
    bbox = bbox.copy()
    if y_flip:
        y_max = H - bbox[:, 0]
        y_min = H - bbox[:, 2]
        bbox[:, 0] = y_min
        bbox[:, 2] = y_max
    if x_flip:
        x_max = W - bbox[:, 1]
        x_min = W - bbox[:, 3]
        bbox[:, 1] = x_min
        bbox[:, 3] = x_max
    return bbox


def _nearest_index(n_out, n_in):
    # cv2 INTER_NEAREST: src = min(floor(dst * (n_in / n_out)), n_in - 1), ratio in double
    idx = np.floor(np.arange(n_out, dtype=np.float64) * (float(n_in) / float(n_out))).astype(np.int64)
    return np.minimum(idx, n_in - 1)


_RESIZE_POOL = None


def resize_nearest(img, size, x_flip=False):
    """chainercv.transforms.resize(img, size, interpolation=0) for a CHW array (cv2
    INTER_NEAREST index rule), optionally followed by a horizontal flip.  Separable: columns are
    gathered first (on the small source), then whole rows are copied; the planes of a mask stack
    are spread over a few threads (NumPy releases the GIL in ``take``) — 34 MB of int32 per
    800 x 1333 COCO image otherwise cost more host time than the GPU needs for the train step."""
    C, H, W = img.shape
    ys = _nearest_index(size[0], H)
    xs = _nearest_index(size[1], W)
    if x_flip:
        xs = xs[::-1]
    out = np.empty((C, size[0], size[1]), dtype=img.dtype)

    def plane(c):
        np.take(np.take(img[c], xs, axis=1), ys, axis=0, out=out[c])

    if C * size[0] * size[1] < (1 << 21) or C == 1:
        for c in range(C):
            plane(c)
    else:
        global _RESIZE_POOL
        if _RESIZE_POOL is None:
            from concurrent.futures import ThreadPoolExecutor
            _RESIZE_POOL = ThreadPoolExecutor(max_workers=4, thread_name_prefix='mrcnn-resize')
        list(_RESIZE_POOL.map(plane, range(C)))
    return out


def flip(img, y_flip=False, x_flip=False):
    """chainercv.transforms.flip for a CHW array."""
    if y_flip:
        img = img[:, ::-1, :]
    if x_flip:
        img = img[:, :, ::-1]
    return img


class MaskRCNNTransform(object):
    """``MaskRCNNTransform(mask_rcnn, train=True)(in_data)`` with ``in_data`` =
    ``(img HWC, bbox, label, mask)`` or the 6-tuple that also carries ``crowd, area``.
    Evaluation mode only transposes the image; training mode returns
    ``(img, bbox, label, mask, scale)``."""

    def __init__(self, mask_rcnn, train=True):
        self.mask_rcnn = mask_rcnn
        self.train = train

    def __call__(self, in_data):
        if len(in_data) not in (4, 6):
            raise ValueError
        img, bbox, label, mask = in_data[:4]
        extras = tuple(in_data[4:])
        chw = img.transpose(2, 0, 1)
        if not self.train:
            return (chw, bbox, label, mask) + extras

        # the reference resizes first and draws the flip afterwards; nothing in between
        # touches an RNG, so drawing first leaves the random stream identical
        x_flip = random.choice([True, False])
        in_size = chw.shape[1:]
        imgs, _, scales = self.mask_rcnn.prepare([chw], x_flips=[x_flip])
        x = imgs[0]                        # device tensor (3, o_H, o_W), channels-last memory
        out_size = tuple(x.shape[1:])

        if len(bbox) > 0:
            bbox = resize_bbox(bbox, in_size, out_size)
        bbox = flip_bbox(bbox, out_size, x_flip=x_flip)

        stack = mask[None] if mask.ndim == 2 else mask
        if len(mask) > 0:
            stack = resize_nearest(stack, out_size, x_flip=x_flip)
        else:
            stack = flip(stack, x_flip=x_flip)
        mask = stack[0] if mask.ndim == 2 else stack
        return x, bbox, label, mask, scales[0]
