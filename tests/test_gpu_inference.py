"""Inference post-processing on the device (decode per class, per-class threshold + sort +
batched NMS) vs the oracle: the detection set is an integer result and must be identical."""
import numpy as np
import pytest
import torch

from oracle import np_infer
import chainer_mask_rcnn_amd as cmr
from chainer_mask_rcnn_amd import _lib

pytestmark = pytest.mark.gpu


def _make(rng, R, n_class, H, W):
    y0 = rng.uniform(0, H - 20, R); x0 = rng.uniform(0, W - 20, R)
    roi = np.stack([y0, x0, np.minimum(y0 + rng.uniform(10, 300, R), H),
                    np.minimum(x0 + rng.uniform(10, 300, R), W)], 1).astype(np.float32)
    loc = (rng.standard_normal((R, n_class * 4)) * 0.5).astype(np.float32)
    logits = (rng.standard_normal((R, n_class)) * 2.5).astype(np.float32)
    return roi, loc, logits


@pytest.mark.parametrize('R,n_class', [(300, 81), (1000, 81), (77, 21)])
def test_detection_set_matches_oracle(dev, R, n_class):
    rng = np.random.RandomState(R)
    H, W = 600, 900
    scale = 1.6
    roi, loc, logits = _make(rng, R, n_class, H * scale, W * scale)
    model = cmr.models.MaskRCNN(None, None, None, mean=None)
    model.head = type('H', (), {'n_class': n_class, 'mask_size': 14})()
    t = lambda a: torch.tensor(a, device=dev)
    # fused-head layout: locs and scores are column slices of one (R, 5*n_class[+pad]) buffer
    width = ((5 * n_class + 3) // 4) * 4
    fc = np.zeros((R, width), np.float32)
    fc[:, :4 * n_class] = loc
    fc[:, 4 * n_class:5 * n_class] = logits
    fct = t(fc)
    bboxes, labels, scores = model._to_bboxes(
        fct[:, :4 * n_class], fct[:, 4 * n_class:5 * n_class], t(roi),
        torch.zeros(R, dtype=torch.int32, device=dev), [(H, W)], [scale])

    prob = cmr.functions.softmax(fct[:, 4 * n_class:5 * n_class]).cpu().numpy()
    cls_bbox = np_infer.decode_cls_boxes(roi, loc, n_class, scale, (H, W))
    b, l, s = np_infer.suppress(cls_bbox, prob, n_class)
    b, l, s = np_infer.finish(b, l, s)
    assert len(bboxes[0]) == len(b)
    assert np.array_equal(labels[0], l)
    assert np.array_equal(scores[0], s)
    assert np.array_equal(bboxes[0], b)


def test_decode_cls_boxes_bit_exact(dev):
    rng = np.random.RandomState(1)
    R, n_class = 500, 81
    roi, loc, _ = _make(rng, R, n_class, 960, 1440)
    out = torch.empty((R, n_class, 4), device=dev)
    mean = (_lib.c_f32 * 4)(0., 0., 0., 0.)
    std = (_lib.c_f32 * 4)(0.1, 0.1, 0.2, 0.2)
    _lib.call('mrcnn_decode_cls_boxes', _lib.ptr(torch.tensor(roi, device=dev)),
              _lib.ptr(torch.tensor(loc, device=dev)), n_class * 4, _lib.ptr(out), R, n_class,
              1.6, mean, std, 600., 900., _lib.stream_ptr())
    ref = np_infer.decode_cls_boxes(roi, loc, n_class, 1.6, (600, 900)).reshape(R, n_class, 4)
    assert (out.cpu().numpy() != ref).mean() < 1e-6
