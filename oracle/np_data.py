"""TEST INFRASTRUCTURE ONLY — NumPy restatement of the reference's input pipeline
(/root/reference/chainer_mask_rcnn/datasets/transforms.py:10-51 and
datasets/concat_examples.py:6-34), for tests/.  The product never imports this module.

"parity unpinned": chainercv (transforms.resize / resize_bbox / random_flip / flip_bbox / flip),
chainer (concat_examples helpers) and cv2 are not installable here; their behaviour is restated
from the published sources (chainercv 0.9-0.13: resize_bbox scales by out/in per axis,
flip_bbox mirrors around the image size, random_flip draws random.choice([True, False]) once
per enabled axis, transforms.resize(interpolation=0) = cv2 INTER_NEAREST when cv2 is present).
"""
import numpy as np

from . import np_infer


def resize_bbox(bbox, in_size, out_size):
    bbox = bbox.copy()
    y_scale = float(out_size[0]) / in_size[0]
    x_scale = float(out_size[1]) / in_size[1]
    for col, sc in ((0, y_scale), (2, y_scale), (1, x_scale), (3, x_scale)):
        bbox[:, col] = sc * bbox[:, col]
    return bbox


def flip_bbox_x(bbox, size):
    W = size[1]
    out = bbox.copy()
    out[:, 1] = W - bbox[:, 3]
    out[:, 3] = W - bbox[:, 1]
    return out


def cv_resize_nearest(img, out_h, out_w):
    """cv2.resize(..., interpolation=INTER_NEAREST) of an (h, w) array: pure-Python index
    loops, literal: sx = min(cvFloor(x * (w / out_w)), w - 1)."""
    h, w = img.shape
    out = np.empty((out_h, out_w), img.dtype)
    fy, fx = float(h) / out_h, float(w) / out_w
    xs = [min(int(np.floor(x * fx)), w - 1) for x in range(out_w)]
    for y in range(out_h):
        sy = min(int(np.floor(y * fy)), h - 1)
        row = img[sy]
        for x in range(out_w):
            out[y, x] = row[xs[x]]
    return out


def transform_train(img_hwc, bbox, label, mask, x_flip, mean, min_size, max_size):
    """transforms.py:22-51 for one example with the flip decision given."""
    img = img_hwc.transpose(2, 0, 1)
    _, H, W = img.shape
    img, scale = np_infer.prepare(img, mean, min_size, max_size)
    _, o_H, o_W = img.shape
    if len(bbox) > 0:
        bbox = resize_bbox(bbox, (H, W), (o_H, o_W))
    if len(mask) > 0:
        mask = np.stack([cv_resize_nearest(m, o_H, o_W) for m in mask])
    if x_flip:
        img = img[:, :, ::-1]
        bbox = flip_bbox_x(bbox, (o_H, o_W))
        mask = mask[:, :, ::-1]
    return img, bbox, label, mask, scale


def concat_padded(arrays, padding):
    """chainer.dataset.convert._concat_arrays with padding: stack into the elementwise-max
    shape, filled with ``padding``."""
    shape = np.max([a.shape for a in arrays], axis=0)
    out = np.full((len(arrays),) + tuple(shape), padding, dtype=arrays[0].dtype)
    for i, a in enumerate(arrays):
        out[(i,) + tuple(slice(0, d) for d in a.shape)] = a
    return out
