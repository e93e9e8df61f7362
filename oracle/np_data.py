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


# --------------------------------------------------------------------------------------
# pycocotools.mask restatement (un-vendored dependency of datasets/coco.py:145-151; algorithm:
# cocoapi common/maskApi.c rleFrString / rleDecode): "parity unpinned".
# --------------------------------------------------------------------------------------

def rle_from_string(s):
    """compressed COCO RLE string -> list of run lengths."""
    if isinstance(s, str):
        s = s.encode('ascii')
    cnts = []
    p = 0
    while p < len(s):
        x = 0
        k = 0
        more = 1
        while more:
            c = s[p] - 48
            x |= (c & 0x1f) << 5 * k
            more = c & 0x20
            p += 1
            k += 1
            if not more and (c & 0x10):
                x |= -1 << 5 * k
        if len(cnts) > 2:
            x += cnts[len(cnts) - 2]
        cnts.append(x)
    return cnts


def rle_to_string(cnts):
    """list of run lengths -> compressed COCO RLE string (maskApi.c rleToString): the inverse,
    used to build test annotations."""
    out = bytearray()
    for i, x in enumerate(cnts):
        x = int(x)
        if i > 2:
            x -= int(cnts[i - 2])
        more = True
        while more:
            c = x & 0x1f
            x >>= 5
            more = (x != -1) if (c & 0x10) else (x != 0)
            if more:
                c |= 0x20
            out.append(c + 48)
    return bytes(out).decode('ascii')


def mask_to_rle_counts(mask):
    """(h, w) {0,1} mask -> uncompressed COCO counts (column-major runs starting with zeros)."""
    flat = np.asarray(mask, np.uint8).T.reshape(-1)
    cnts, v, run = [], 0, 0
    for b in flat:
        if b != v:
            cnts.append(run)
            run, v = 0, b
        run += 1
    cnts.append(run)
    return cnts


def rle_decode(rle):
    """{'counts': list | str, 'size': [h, w]} -> (h, w) uint8 (pycocotools.mask.decode)."""
    h, w = rle['size']
    cnts = rle['counts'] if isinstance(rle['counts'], (list, tuple)) else rle_from_string(rle['counts'])
    out = np.zeros(h * w, np.uint8)
    pos, v = 0, 0
    for c in cnts:
        for _ in range(c):
            out[pos] = v
            pos += 1
        v = 1 - v
    return out.reshape(w, h).transpose(1, 0)
