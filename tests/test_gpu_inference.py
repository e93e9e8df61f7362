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
    import ctypes
    mean = (ctypes.c_double * 4)(0., 0., 0., 0.)
    std = (ctypes.c_double * 4)(0.1, 0.1, 0.2, 0.2)
    roi_d, loc_d = torch.tensor(roi, device=dev), torch.tensor(loc, device=dev)   # keep alive
    _lib.call('mrcnn_decode_cls_boxes', _lib.ptr(roi_d), _lib.ptr(loc_d), n_class * 4,
              _lib.ptr(out), R, n_class, 1.6, mean, std, 600., 900., _lib.stream_ptr())
    ref = np_infer.decode_cls_boxes(roi, loc, n_class, 1.6, (600, 900)).reshape(R, n_class, 4)
    assert np.array_equal(out.cpu().numpy(), ref)


def test_prepare_matches_oracle(dev):
    """MaskRCNN.prepare on the device (resize + mean + zero-padded batch) vs the oracle."""
    torch.manual_seed(0)
    rng = np.random.RandomState(2)
    model = cmr.models.MaskRCNNResNet(50, n_fg_class=80, min_size=160, max_size=240,
                                      anchor_scales=(2, 4, 8, 16, 32), roi_size=14).to(dev)
    imgs = [rng.randint(0, 256, (3, 97, 131)).astype(np.uint8),
            rng.uniform(0, 255, (3, 120, 90)).astype(np.float32),
            rng.randint(0, 256, (3, 60, 200)).astype(np.uint8)]      # max_size rule kicks in
    x, sizes, scales = model.prepare(imgs)
    x = x.cpu().numpy()
    assert sizes == [(97, 131), (120, 90), (60, 200)]
    for n, img in enumerate(imgs):
        ref, scale = np_infer.prepare(img, model.mean.ravel(), 160, 240)
        assert abs(scale - scales[n]) < 1e-12
        h, w = ref.shape[1:]
        np.testing.assert_allclose(x[n, :, :h, :w], ref, rtol=0, atol=2e-4)
        assert (x[n, :, h:, :] == 0).all() and (x[n, :, :, w:] == 0).all()   # padding=0


def test_paste_masks_matches_oracle(dev):
    """segm_results on the device vs the oracle: the pasted image-size masks are an integer
    (bool) result."""
    rng = np.random.RandomState(3)
    D, n_fg, M, im_h, im_w = 40, 80, 14, 150, 210
    logits = (rng.standard_normal((D, n_fg, M, M)) * 3).astype(np.float32)
    label = rng.randint(0, n_fg, D).astype(np.int32)
    y0 = rng.uniform(-10, im_h - 5, D); x0 = rng.uniform(-10, im_w - 5, D)
    bbox = np.stack([y0, x0, y0 + rng.uniform(1, 120, D), x0 + rng.uniform(1, 150, D)], 1).astype(np.float32)
    bbox[0] = [3.2, 4.7, 3.9, 5.1]                      # sub-pixel box
    bbox[1] = [-20, -30, im_h + 15, im_w + 40]          # larger than the image
    model = cmr.models.MaskRCNN(None, None, None, mean=None)
    masks = model._to_masks([bbox], [label], None, [torch.tensor(logits, device=dev)], [(im_h, im_w)])
    ref = np_infer.segm_results(bbox, label, logits, im_h, im_w)
    assert masks[0].dtype == bool and masks[0].shape == ref.shape
    assert np.array_equal(masks[0], ref)
    assert ref.any()


def test_predict_end_to_end_api(dev):
    """model.predict(list of CHW images) -> (bboxes, masks, labels, scores) as the reference."""
    torch.manual_seed(0)
    rng = np.random.RandomState(4)
    model = cmr.models.MaskRCNNResNet(50, n_fg_class=80, min_size=160, max_size=240,
                                      anchor_scales=(2, 4, 8, 16, 32), roi_size=14,
                                      proposal_creator_params=dict(min_size=0, n_test_pre_nms=300,
                                                                   n_test_post_nms=50)).to(dev)
    with torch.no_grad():
        model.extractor.bn1.W.fill_(1. / 64.)
        model.head.cls_loc_score.W[4 * 81:5 * 81] *= 300.
    imgs = [rng.randint(0, 256, (3, 100, 140)).astype(np.uint8),
            rng.randint(0, 256, (3, 120, 90)).astype(np.uint8)]
    bboxes, masks, labels, scores = model.predict(imgs)
    assert len(bboxes) == len(masks) == len(labels) == len(scores) == 2
    for img, b, m, l, s in zip(imgs, bboxes, masks, labels, scores):
        assert b.dtype == np.float32 and l.dtype == np.int32 and s.dtype == np.float32
        assert m.dtype == bool and m.shape == (len(b),) + img.shape[1:]
        assert len(b) == len(l) == len(s) <= 100
        if len(b):
            assert b[:, 0::2].min() >= 0 and b[:, 0::2].max() <= img.shape[1]
            assert (s > 0.05).all() and l.min() >= 0 and l.max() < 80


def test_to_bboxes_matches_reference_method_fixture(dev, golden_dir):
    """The device path of MaskRCNN._to_bboxes (softmax, per-class decode, batched threshold /
    sort / NMS, host finish) on the inputs of tests/golden/to_bboxes.npz, whose outputs come
    from the reference's own method bodies (oracle/gen_golden.py section 7)."""
    import os
    d = np.load(os.path.join(golden_dir, 'to_bboxes.npz'))
    n_class = d['probs'].shape[1]
    model = cmr.models.MaskRCNN(None, None, None, mean=None)
    model.head = type('H', (), {'n_class': n_class, 'mask_size': 14})()
    t = lambda a: torch.tensor(a, device=dev)
    bboxes, labels, scores = model._to_bboxes(
        t(d['roi_cls_locs']), t(d['roi_scores']), t(d['rois']), t(d['roi_indices']),
        [tuple(int(v) for v in s) for s in d['sizes']], [float(s) for s in d['scales']])
    lo = 0
    for i, n in enumerate(d['n_det']):
        assert len(bboxes[i]) == n
        assert np.array_equal(labels[i], d['label'][lo:lo + n])
        assert np.array_equal(bboxes[i], d['bbox'][lo:lo + n])
        np.testing.assert_allclose(scores[i], d['score'][lo:lo + n], rtol=2e-6, atol=0)
        lo += n


def test_image_io_matches_reference_body_fixtures(dev, golden_dir):
    """Device prepare / mask paste vs the fixtures produced by the reference's own
    `MaskRCNN.prepare` and `segm_results` bodies (oracle/gen_golden.py section 9)."""
    import os
    from test_oracle_boxes import _segm_fixture
    d = np.load(os.path.join(golden_dir, 'prepare.npz'))
    model = cmr.models.MaskRCNNResNet(50, n_fg_class=80, min_size=int(d['min_size']),
                                      max_size=int(d['max_size']), mean=tuple(d['mean']),
                                      anchor_scales=(2, 4, 8, 16, 32), roi_size=14).to(dev)
    x, sizes, scales = model.prepare([d['img%d' % i] for i in range(3)])
    x = x.cpu().numpy()
    for i in range(3):
        ref = d['out%d' % i]
        assert scales[i] == float(d['scales'][i]) and tuple(sizes[i]) == tuple(d['sizes'][i])
        np.testing.assert_allclose(x[i, :, :ref.shape[1], :ref.shape[2]], ref, rtol=0, atol=2e-4)
    bbox, label, logits, im_h, im_w, masks = _segm_fixture(golden_dir)
    got = model._to_masks([bbox], [label], None, [torch.tensor(logits, device=dev)], [(im_h, im_w)])
    assert got[0].shape == masks.shape and np.array_equal(got[0], masks)


def test_predict_with_no_detections(dev):
    """Nothing above the score threshold: empty per-image results of the right types/shapes
    (the reference returns empty arrays, mask_rcnn.py:63-65 for the masks)."""
    torch.manual_seed(0)
    rng = np.random.RandomState(9)
    model = cmr.models.MaskRCNNResNet(50, n_fg_class=80, min_size=128, max_size=192,
                                      anchor_scales=(2, 4, 8, 16, 32), roi_size=14,
                                      proposal_creator_params=dict(min_size=0, n_test_pre_nms=200,
                                                                   n_test_post_nms=30)).to(dev)
    model.score_thresh = 1.1
    imgs = [rng.randint(0, 256, (3, 90, 120)).astype(np.uint8),
            rng.randint(0, 256, (3, 100, 80)).astype(np.uint8)]
    bboxes, masks, labels, scores = model.predict(imgs)
    for img, b, m, l, s in zip(imgs, bboxes, masks, labels, scores):
        assert b.shape == (0, 4) and b.dtype == np.float32
        assert l.shape == (0,) and l.dtype == np.int32 and s.shape == (0,) and s.dtype == np.float32
        assert m.shape == (0,) + img.shape[1:] and m.dtype == bool


def test_c5_full_size_predict(dev):
    """BASELINE configs[4] at FULL size: ResNet50-C4 inference on 8 x 3 x 1024 x 1024 with
    n_test_pre_nms 6000 / n_test_post_nms 1000 (models/mask_rcnn_resnet.py:48-52), per-class
    NMS and the mask head on the <= 100 detections per image (models/mask_rcnn.py:307-337).
    The detection sets (integer decisions: which RoI/class pairs survive) must equal the
    oracle's `_to_bboxes` fed with the HIP head outputs."""
    torch.manual_seed(0)
    rng = np.random.RandomState(0)
    N, H, W = 8, 1024, 1024
    model = cmr.models.MaskRCNNResNet(50, n_fg_class=80, min_size=800, max_size=1333,
                                      anchor_scales=(2, 4, 8, 16, 32), roi_size=14).to(dev)
    from chainer_mask_rcnn_amd.models.resnet_extractor import Bottleneck
    with torch.no_grad():
        # no trained weights offline: keep activations O(1) and sharpen the class scores so that
        # the synthetic run produces detections (same recipe as bench.py --workload infer)
        model.extractor.bn1.W.fill_(1. / 64.)
        for m in model.modules():
            if isinstance(m, Bottleneck):
                m.bn3.W.fill_(0.25)
                if m.projection:
                    m.bn4.W.fill_(0.5)
        model.head.cls_loc_score.W[4 * 81:5 * 81] *= 60.
    mean = np.asarray(model.mean, np.float32).reshape(3, 1, 1)
    x = torch.tensor(rng.uniform(0, 255, (N, 3, H, W)).astype(np.float32) - mean, device=dev)
    scales, sizes = [1.6] * N, [(640, 640)] * N
    bboxes, roi_masks, labels, scores, mid = model.predict_prepared(
        x, scales, sizes, return_intermediates=True)
    assert mid['feature_shape'] == (N, 1024, 65, 65)               # 1024 -> 512 -> 257 -> 129 -> 65
    idx = mid['roi_indices'].cpu().numpy()
    counts = np.bincount(idx, minlength=N)
    assert (counts <= 1000).all() and counts.sum() > 0
    assert (np.diff(idx) >= 0).all()                                # grouped by image, in order
    n_class = 81
    rois = mid['rois'].cpu().numpy()
    locs = mid['roi_cls_locs'].cpu().numpy()
    probs = cmr.functions.softmax(mid['roi_scores']).cpu().numpy()
    total = 0
    for i in range(N):
        sel = idx == i
        cls_bbox = np_infer.decode_cls_boxes(rois[sel], np.ascontiguousarray(locs[sel]), n_class,
                                             scales[i], sizes[i])
        b, l, s = np_infer.suppress(cls_bbox, probs[sel], n_class)
        b, l, s = np_infer.finish(b, l, s)
        assert len(bboxes[i]) == len(b) <= 100
        assert np.array_equal(labels[i], l)
        assert np.array_equal(bboxes[i], b)
        assert np.array_equal(scores[i], s)
        assert roi_masks[i].shape == (len(b), 80, 14, 14) and np.isfinite(roi_masks[i]).all()
        total += len(b)
    assert total > 0, 'synthetic weights produced no detections: the NMS / mask stages did not run'
    print('C5: proposals/img', counts.tolist(), 'detections/img', [len(b) for b in bboxes])
