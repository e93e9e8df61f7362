"""Input pipeline: host logic of MaskRCNNTransform / concat_examples vs the
oracle restatement; the image half needs the device and is covered in test_gpu_datasets.py."""
import random

import numpy as np
import pytest
import torch

from chainer_mask_rcnn_amd import datasets as D
from oracle import np_data


class _StubModel(object):
    """Stands in for MaskRCNN.prepare on a CPU-only box: same scale rule, zero image."""
    min_size, max_size = 100, 150

    def __init__(self):
        self.calls = []

    def prepare(self, imgs, x_flips=None):
        self.calls.append(list(x_flips))
        _, H, W = imgs[0].shape
        scale = self.min_size / min(H, W)
        if scale * max(H, W) > self.max_size:
            scale = self.max_size / max(H, W)
        oh, ow = int(np.round(H * scale)), int(np.round(W * scale))
        return [torch.zeros((3, oh, ow))], [(H, W)], [scale]


def _example(rng, H=61, W=83, G=3):
    img = rng.randint(0, 256, (H, W, 3)).astype(np.uint8)
    y0 = rng.uniform(0, H / 2, G); x0 = rng.uniform(0, W / 2, G)
    bbox = np.stack([y0, x0, y0 + rng.uniform(4, H / 2, G), x0 + rng.uniform(4, W / 2, G)], 1).astype(np.float32)
    label = rng.randint(0, 80, G).astype(np.int32)
    mask = (rng.uniform(size=(G, H, W)) > 0.5).astype(np.int32)
    return img, bbox, label, mask


def test_box_and_mask_helpers_match_oracle():
    rng = np.random.RandomState(0)
    _, bbox, _, mask = _example(rng)
    assert np.array_equal(D.resize_bbox(bbox, (61, 83), (100, 136)), np_data.resize_bbox(bbox, (61, 83), (100, 136)))
    assert np.array_equal(D.flip_bbox(bbox, (100, 136), x_flip=True), np_data.flip_bbox_x(bbox, (100, 136)))
    for size in [(100, 136), (30, 200), (61, 83), (7, 5)]:
        ref = np.stack([np_data.cv_resize_nearest(m, *size) for m in mask])
        assert np.array_equal(D.resize_nearest(mask, size), ref)
        assert np.array_equal(D.resize_nearest(mask, size, x_flip=True), ref[:, :, ::-1])
    assert np.array_equal(D.flip(mask, x_flip=True), mask[:, :, ::-1])


@pytest.mark.parametrize('seed', [0, 1, 2, 3])
def test_transform_targets_match_oracle_and_rng_stream(seed):
    rng = np.random.RandomState(seed)
    ex = _example(rng)
    model = _StubModel()
    random.seed(seed)
    img, bbox, label, mask, scale = D.MaskRCNNTransform(model)(ex)
    after = random.random()
    random.seed(seed)
    x_flip = random.choice([True, False])       # what chainercv's random_flip draws
    assert after == random.random()             # exactly one draw consumed
    assert model.calls == [[x_flip]]
    _, rb, rl, rm, rs = np_data.transform_train(ex[0], ex[1], ex[2], ex[3], x_flip,
                                                (0., 0., 0.), model.min_size, model.max_size)
    assert scale == rs and np.array_equal(bbox, rb) and np.array_equal(label, rl)
    assert mask.shape == rm.shape and np.array_equal(mask, rm)
    assert tuple(img.shape[1:]) == mask.shape[1:]


def test_transform_eval_mode_and_errors():
    rng = np.random.RandomState(5)
    ex = _example(rng)
    t = D.MaskRCNNTransform(_StubModel(), train=False)
    out = t(ex)
    assert len(out) == 4 and out[0].shape == (3, 61, 83) and out[1] is ex[1]
    out6 = t(ex + (np.zeros(3, bool), np.ones(3, np.float32)))
    assert len(out6) == 6
    with pytest.raises(ValueError):
        t(ex[:3])
    # empty ground truth passes through
    img, bbox, label, mask, scale = D.MaskRCNNTransform(_StubModel())(
        (ex[0], np.zeros((0, 4), np.float32), np.zeros((0,), np.int32), np.zeros((0, 61, 83), np.int32)))
    assert bbox.shape == (0, 4) and mask.shape[0] == 0


def test_concat_examples_semantics():
    rng = np.random.RandomState(1)
    a = (rng.rand(3, 5, 7).astype(np.float32), rng.rand(2, 4).astype(np.float32),
         np.array([1, 2], np.int32), np.ones((2, 5, 7), np.int32), np.float32(1.5))
    b = (rng.rand(3, 6, 4).astype(np.float32), rng.rand(3, 4).astype(np.float32),
         np.array([3, 4, 5], np.int32), np.ones((3, 6, 4), np.int32), np.float32(0.5))
    # the train converter of examples/train_common.py:219-225
    imgs, bboxes, labels, masks, scales = D.concat_examples(
        [a, b], None, padding=0, indices_concat=[0, 2, 3, 4], indices_to_device=[0, 1])
    assert np.array_equal(imgs, np_data.concat_padded([a[0], b[0]], 0)) and imgs.shape == (2, 3, 6, 7)
    assert isinstance(bboxes, list) and bboxes[0] is a[1]
    assert np.array_equal(labels, [[1, 2, 0], [3, 4, 5]])
    assert masks.shape == (2, 3, 6, 7) and masks[0, 2].sum() == 0
    assert np.array_equal(scales, [1.5, 0.5])
    # device placement: listed entries become tensors, the rest stays NumPy
    out = D.concat_examples([a, b], torch.device('cpu'), padding=0,
                            indices_concat=[0, 2, 3, 4], indices_to_device=[0, 1])
    assert isinstance(out[0], torch.Tensor) and isinstance(out[1][0], torch.Tensor)
    assert isinstance(out[3], np.ndarray)
    # tensors from the transform are padded on their device in channels-last memory
    ta, tb = torch.rand(3, 5, 7), torch.rand(3, 6, 4)
    x, = D.concat_examples([(ta,), (tb,)], None, padding=0)
    assert x.shape == (2, 3, 6, 7) and x.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(x[0, :, :5, :7], ta) and float(x[0, :, 5:, :].abs().sum()) == 0.
    # no padding: plain stack; per-field padding tuple; empty batch
    s, = D.concat_examples([(a[0],), (a[0],)])
    assert s.shape == (2, 3, 5, 7)
    p = D.concat_examples([a[:2], b[:2]], padding=(0, -1))
    assert p[1].shape == (2, 3, 4) and p[1][0, 2, 0] == -1
    with pytest.raises(ValueError):
        D.concat_examples([])
