"""tools/train_loop.py on the device: dataset (a COCO-layout directory written here: JPEGs +
polygon / RLE annotations) -> MaskRCNNTransform -> concat_examples -> optimizer.update, the input
pipeline one batch ahead on its own thread and stream (examples/train_common.py:200-231).

* the loop's first step is the direct call on the hand-assembled first batch (same loss bits);
* the prefetching loop and the serial (reference-order) loop give bit-identical losses over an
  epoch boundary, and leave Python's `random` and the global `np.random` stream at the same place."""
import json
import os
import random
import sys

import numpy as np
import pytest
import torch

import chainer_mask_rcnn_amd as cmr
from chainer_mask_rcnn_amd import optimizers

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
import train_loop as TL

pytestmark = pytest.mark.gpu


def _write_coco(root, n_images=5, H=96, W=128):
    import PIL.Image
    rng = np.random.RandomState(0)
    os.makedirs(os.path.join(root, 'annotations'))
    os.makedirs(os.path.join(root, 'val2014'))
    images, anns = [], []
    for i in range(n_images):
        img_id = 11 + i
        img = rng.randint(0, 256, (H, W, 3)).astype(np.uint8)
        PIL.Image.fromarray(img).save(os.path.join(root, 'val2014', 'COCO_val2014_%012d.jpg' % img_id),
                                      quality=95)
        images.append(dict(id=img_id, height=H, width=W))
        for g in range(1 + i % 3):
            y0, x0 = int(rng.randint(4, H // 2)), int(rng.randint(4, W // 2))
            h, w = int(rng.randint(24, H // 2 - 4)), int(rng.randint(24, W // 2 - 4))
            poly = [x0, y0, x0 + w, y0, x0 + w, y0 + h, x0 + w // 2, y0 + h + 3, x0, y0 + h]
            anns.append(dict(id=len(anns) + 1, image_id=img_id, category_id=[1, 3, 18][g],
                             segmentation=[[float(v) for v in poly]], iscrowd=0, area=float(h * w),
                             bbox=[x0, y0, w, h]))
    cats = [dict(id=c, name='cat%d' % c) for c in (1, 3, 18)]
    with open(os.path.join(root, 'annotations', 'instances_minival2014.json'), 'w') as f:
        json.dump(dict(images=images, annotations=anns, categories=cats), f)


def _build(dev, data, prefetch, seed=4):
    random.seed(seed)                     # examples/train_common.py:135-136
    np.random.seed(seed)
    torch.manual_seed(seed)
    model = cmr.models.MaskRCNNResNet(
        50, n_fg_class=80, anchor_scales=(2, 4, 8, 16, 32), roi_size=14, min_size=144, max_size=192,
        proposal_creator_params=dict(min_size=0, n_train_pre_nms=600, n_train_post_nms=100))
    chain = cmr.models.MaskRCNNTrainChain(
        model, proposal_target_creator=cmr.models.utils.ProposalTargetCreator(n_sample=32)).to(dev)
    chain.train()
    opt = optimizers.MomentumSGD(lr=0.0025, momentum=0.9)
    opt.setup(chain)
    opt.add_hook(optimizers.WeightDecay(1e-4))
    optimizers.disable_update(model.extractor.conv1)
    optimizers.disable_update(model.extractor.bn1)
    optimizers.disable_update(model.extractor.res2)
    for m in chain.modules():
        if isinstance(m, cmr.links.AffineChannel2D):
            optimizers.disable_update(m)
    with torch.no_grad():
        model.extractor.bn1.W.fill_(1. / 64.)
        for m in model.modules():
            if isinstance(m, cmr.models.resnet_extractor.Bottleneck):
                m.bn3.W.fill_(0.25)
    train = TL.TransformDataset(data, cmr.datasets.MaskRCNNTransform(model))
    it = TL.SerialIterator(train, 2)
    return TL.TrainLoop(it, chain, opt, dev, prefetch=prefetch), model, chain, opt, train


def test_loop_over_a_coco_directory(dev, tmp_path):
    root = os.path.join(str(tmp_path), 'COCO')
    _write_coco(root)
    data = cmr.datasets.COCOInstanceSegmentationDataset('minival', root_dir=root)
    assert len(data) == 5

    # (1) first step == the direct call on the hand-assembled first batch
    loop, model, chain, opt, train = _build(dev, data, prefetch=True)
    first = loop.step()
    rep_loop = {k: float(v) for k, v in chain.report.items()}
    loop.close()
    random.seed(4); np.random.seed(4); torch.manual_seed(4)
    loop2, model2, chain2, opt2, train2 = _build(dev, data, prefetch=False)
    order = loop2.iterator._order.copy()                    # drawn by the iterator's constructor
    batch = TL.make_converter(dev)([train2[int(j)] for j in order[:2]])
    imgs, bboxes, labels, masks, scales = batch
    assert imgs.is_cuda and imgs.shape[0] == 2 and imgs.shape[1] == 3
    assert isinstance(bboxes, list) and isinstance(labels, np.ndarray) and masks.dtype == np.int32
    direct = opt2.update(chain2, imgs, bboxes, labels, masks, scales)
    torch.cuda.synchronize()
    assert float(first.detach()) == float(direct.detach())
    assert rep_loop == {k: float(v) for k, v in chain2.report.items()}

    # (2) prefetching == serial, across the epoch boundary (5 images, batch 2: third batch wraps)
    runs = []
    for prefetch in (True, False):
        loop, model, chain, opt, _ = _build(dev, data, prefetch=prefetch)
        losses = [float(l.detach()) for l in loop.run(4)]
        opt.flush()
        torch.cuda.synchronize()
        loop.close()
        w = model.head.res5.a.conv1.W.detach().cpu().numpy().copy()
        runs.append((losses, loop.iterator.epoch, random.random(), np.random.randint(0, 1 << 30), w))
    assert runs[0][0] == runs[1][0] and all(np.isfinite(runs[0][0]))
    assert runs[0][1] == runs[1][1] == 1
    assert runs[0][2] == runs[1][2] and runs[0][3] == runs[1][3]      # both random streams
    assert np.array_equal(runs[0][4], runs[1][4])
