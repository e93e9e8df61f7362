"""Input pipeline on the device: MaskRCNNTransform + concat_examples with the
real MaskRCNN.prepare kernel vs the oracle restatement of the reference's transform."""
import random

import numpy as np
import pytest
import torch

import chainer_mask_rcnn_amd as cmr
from chainer_mask_rcnn_amd import datasets as D
from oracle import np_data

pytestmark = pytest.mark.gpu


def _example(rng, H, W, G, dtype=np.uint8):
    img = rng.randint(0, 256, (H, W, 3)).astype(dtype)
    y0 = rng.uniform(0, H / 2, G); x0 = rng.uniform(0, W / 2, G)
    bbox = np.stack([y0, x0, y0 + rng.uniform(4, H / 2, G), x0 + rng.uniform(4, W / 2, G)], 1).astype(np.float32)
    label = rng.randint(0, 80, G).astype(np.int32)
    mask = (rng.uniform(size=(G, H, W)) > 0.5).astype(np.int32)
    return img, bbox, label, mask


def test_transform_and_converter_match_oracle(dev):
    torch.manual_seed(0)
    model = cmr.models.MaskRCNNResNet(50, n_fg_class=80, min_size=160, max_size=240,
                                      anchor_scales=(2, 4, 8, 16, 32), roi_size=14).to(dev)
    transform = D.MaskRCNNTransform(model)
    rng = np.random.RandomState(4)
    examples = [_example(rng, 97, 131, 3), _example(rng, 120, 90, 2, np.float32),
                _example(rng, 60, 200, 4), _example(rng, 97, 131, 1)]
    random.seed(7)
    flips = [random.choice([True, False]) for _ in examples]
    assert True in flips and False in flips
    random.seed(7)
    outs = [transform(ex) for ex in examples]
    refs = [np_data.transform_train(ex[0], ex[1], ex[2], ex[3], f, model.mean.ravel(), 160, 240)
            for ex, f in zip(examples, flips)]
    for (img, bbox, label, mask, scale), (ri, rb, rl, rm, rs) in zip(outs, refs):
        assert img.is_cuda and tuple(img.shape) == ri.shape
        # uint8 sources are uploaded as bytes and converted in the kernel: same values
        np.testing.assert_allclose(img.cpu().numpy(), ri, rtol=0, atol=2e-4)
        assert abs(scale - rs) < 1e-12
        assert np.array_equal(bbox, rb) and np.array_equal(label, rl) and np.array_equal(mask, rm)

    # the train converter (examples/train_common.py:219-225)
    imgs, bboxes, labels, masks, scales = D.concat_examples(
        outs, dev, padding=0, indices_concat=[0, 2, 3, 4], indices_to_device=[0, 1])
    Hm = max(r[0].shape[1] for r in refs); Wm = max(r[0].shape[2] for r in refs)
    assert imgs.is_cuda and tuple(imgs.shape) == (4, 3, Hm, Wm)
    assert imgs.is_contiguous(memory_format=torch.channels_last)
    ref_batch = np_data.concat_padded([r[0] for r in refs], 0)
    np.testing.assert_allclose(imgs.cpu().numpy(), ref_batch, rtol=0, atol=2e-4)
    assert all(b.is_cuda for b in bboxes) and isinstance(masks, np.ndarray)
    assert np.array_equal(masks, np_data.concat_padded([r[3] for r in refs], 0))
    assert np.array_equal(labels, np_data.concat_padded([r[2] for r in refs], 0))

    # and the batch drives a train step as it is
    chain = cmr.models.MaskRCNNTrainChain(model)
    G = [len(r[1]) for r in refs]
    loss = chain(imgs[:2], [b.cpu().numpy() for b in bboxes[:2]],
                 [labels[i, :G[i]] for i in range(2)], [masks[i, :G[i]] for i in range(2)],
                 [float(s) for s in scales[:2]])
    assert torch.isfinite(loss)
