"""Known-answer tests pinning the NumPy/C restatements of the chainercv box
utilities (no reference golden vectors exist for them: "parity unpinned")."""
import numpy as np
import pytest

import oracle
from oracle import np_ref


def _rand_boxes(rng, n, size=800.):
    cy, cx = rng.uniform(0, size, n), rng.uniform(0, size, n)
    h, w = rng.uniform(4, 300, n), rng.uniform(4, 300, n)
    b = np.stack([cy - h / 2, cx - w / 2, cy + h / 2, cx + w / 2], 1)
    return np.clip(b, 0, size).astype(np.float32)


def _iou_def(a, b):
    ih = max(0., min(a[2], b[2]) - max(a[0], b[0]))
    iw = max(0., min(a[3], b[3]) - max(a[1], b[1]))
    inter = ih * iw
    ua = (a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - inter
    return inter / ua if ua > 0 else 0.


def _nms_brute(bbox, thresh):
    keep = []
    for i in range(len(bbox)):
        if all(_iou_def(bbox[i].astype(np.float64), bbox[k].astype(np.float64)) < thresh
               for k in keep):
            keep.append(i)
    return np.array(keep, np.int32)


def test_bbox_iou_matches_definition():
    rng = np.random.RandomState(0)
    a, b = _rand_boxes(rng, 40), _rand_boxes(rng, 30)
    iou = np_ref.bbox_iou(a, b)
    ref = np.array([[_iou_def(x, y) for y in b] for x in a])
    np.testing.assert_allclose(iou, ref, rtol=1e-5, atol=1e-6)


def test_loc2bbox_bbox2loc_roundtrip():
    rng = np.random.RandomState(1)
    src, dst = _rand_boxes(rng, 50), _rand_boxes(rng, 50)
    loc = np_ref.bbox2loc(src, dst)
    back = np_ref.loc2bbox(src, loc)
    np.testing.assert_allclose(back, dst, rtol=1e-4, atol=1e-2)


def test_anchor_base_coco():
    ab = np_ref.generate_anchor_base(16, (0.5, 1, 2), (2, 4, 8, 16, 32))
    assert ab.shape == (15, 4) and ab.dtype == np.float32
    # ratio 1, scale 8 -> 128x128 centred on (8, 8)
    np.testing.assert_allclose(ab[1 * 5 + 2], [8 - 64, 8 - 64, 8 + 64, 8 + 64])
    anchors = np_ref.enumerate_shifted_anchor(ab, 16, 51, 84)
    assert anchors.shape == (64260, 4)
    np.testing.assert_allclose(anchors[15 + 7] - anchors[7], [0, 16, 0, 16])


@pytest.mark.parametrize('thresh', [0.3, 0.5, 0.7])
def test_nms_numpy_c_and_bruteforce_agree(thresh):
    rng = np.random.RandomState(2)
    bbox = _rand_boxes(rng, 400)
    k_np = np_ref.non_maximum_suppression(bbox, thresh)
    k_c = oracle.nms_sorted(bbox, thresh)
    k_bf = _nms_brute(bbox, thresh)
    assert (k_np == k_c).all()
    assert (k_np == k_bf).all()


def test_nms_with_score_and_limit():
    rng = np.random.RandomState(3)
    bbox = _rand_boxes(rng, 200)
    score = rng.permutation(200).astype(np.float32)
    k = np_ref.non_maximum_suppression(bbox, 0.5, score=score, limit=7)
    assert len(k) == 7 and k.dtype == np.int32
    assert (np.diff(score[k]) < 0).all()
    assert len(np_ref.non_maximum_suppression(np.zeros((0, 4), np.float32), 0.5)) == 0


def test_nms_degenerate_boxes():
    # zero-area boxes give iou = 0/0 = NaN -> never suppress (A.3)
    bbox = np.array([[5, 5, 5, 5], [5, 5, 5, 5], [0, 0, 10, 10], [0, 0, 10, 10]], np.float32)
    assert list(np_ref.non_maximum_suppression(bbox, 0.5)) == [0, 1, 2]
    assert list(oracle.nms_sorted(bbox, 0.5)) == [0, 1, 2]


def test_proposal_creator_counts_and_order():
    rng = np.random.RandomState(4)
    ab = np_ref.generate_anchor_base(16, (0.5, 1, 2), (2, 4, 8, 16, 32))
    anchor = np_ref.enumerate_shifted_anchor(ab, 16, 12, 20)
    loc = (rng.standard_normal((len(anchor), 4)) * 0.1).astype(np.float32)
    score = rng.standard_normal(len(anchor)).astype(np.float32)
    pc = np_ref.ProposalCreator(min_size=0, n_train_pre_nms=600, n_train_post_nms=50,
                                n_test_pre_nms=300, n_test_post_nms=20)
    roi, idx = pc(loc, score, anchor, (192, 320), 1., train=True, return_indices=True)
    assert roi.shape[0] <= 50 and roi.shape[1] == 4
    assert (np.diff(score[idx]) <= 0).all()
    assert roi[:, 0::2].min() >= 0 and roi[:, 0::2].max() <= 192
    roi_t = pc(loc, score, anchor, (192, 320), 1., train=False)
    assert roi_t.shape[0] <= 20


def test_anchor_target_creator_properties():
    rng = np.random.RandomState(5)
    np.random.seed(0)
    ab = np_ref.generate_anchor_base(16, (0.5, 1, 2), (2, 4, 8, 16, 32))
    anchor = np_ref.enumerate_shifted_anchor(ab, 16, 20, 30)
    bbox = np.array([[30, 40, 200, 260], [100, 300, 250, 420]], np.float32)
    loc, label = np_ref.AnchorTargetCreator()(bbox, anchor, (320, 480))
    assert loc.shape == (len(anchor), 4) and label.shape == (len(anchor),)
    assert set(np.unique(label)) <= {-1, 0, 1}
    assert (label == 1).sum() <= 128 and (label >= 0).sum() <= 256
    assert (label == 1).sum() >= 2   # at least the per-gt argmax anchors


def test_resize_bilinear_identity_and_constant():
    img = np.arange(20, dtype=np.float32).reshape(4, 5)
    np.testing.assert_allclose(np_ref.resize_bilinear(img, 4, 5), img)
    const = np.full((9, 7), 3.5, np.float32)
    np.testing.assert_allclose(np_ref.resize_bilinear(const, 14, 14), 3.5)
    up = np_ref.resize_bilinear(np.array([[0., 1.]], np.float32), 1, 4)
    np.testing.assert_allclose(up, [[0., 0.25, 0.75, 1.]])


def test_shifted_anchor_and_expand_boxes_pinned_to_reference(golden_dir):
    """Golden vectors produced by the reference's own `_enumerate_shifted_anchor`
    (models/region_proposal_network.py:148-167) and `expand_boxes` (models/mask_rcnn.py:44-60):
    the oracle and the product's host helper reproduce them exactly."""
    import os
    from chainer_mask_rcnn_amd.utils import bbox as pb
    from oracle import np_infer
    d = np.load(os.path.join(golden_dir, 'shifted_anchor.npz'))
    base, stride = d['anchor_base'], int(d['feat_stride'])
    assert np.array_equal(pb.generate_anchor_base(16, (0.5, 1, 2), (2, 4, 8, 16, 32)), base)
    for k, (h, w) in enumerate(d['hw']):
        ref = d['a%d' % k]
        assert ref.dtype == np.float32 and ref.shape == (h * w * len(base), 4)
        assert np.array_equal(np_ref.enumerate_shifted_anchor(base, stride, int(h), int(w)), ref)
        assert np.array_equal(pb.enumerate_shifted_anchor(base, stride, int(h), int(w)), ref)
    e = np.load(os.path.join(golden_dir, 'expand_boxes.npz'))
    out = np_infer.expand_boxes(e['boxes'], float(e['scale']))
    assert out.dtype == e['out'].dtype and np.array_equal(out, e['out'])


def test_inference_postprocessing_pinned_to_reference_methods(golden_dir):
    """Fixture produced by the reference's own `MaskRCNN._to_bboxes` / `_suppress` bodies
    (models/mask_rcnn.py:178-265) on top of restated loc2bbox / NMS (oracle/gen_golden.py
    section 7): the oracle's decode -> suppress -> finish pipeline reproduces it exactly."""
    import os
    from oracle import np_infer
    d = np.load(os.path.join(golden_dir, 'to_bboxes.npz'))
    n_class = d['probs'].shape[1]
    lo = 0
    for i, n in enumerate(d['n_det']):
        sel = d['roi_indices'] == i
        cls_bbox = np_infer.decode_cls_boxes(d['rois'][sel], d['roi_cls_locs'][sel], n_class,
                                             float(d['scales'][i]), tuple(d['sizes'][i]))
        b, l, s = np_infer.finish(*np_infer.suppress(cls_bbox, d['probs'][sel], n_class))
        assert len(b) == n
        assert np.array_equal(b, d['bbox'][lo:lo + n])
        assert l.dtype == np.int32 and np.array_equal(l, d['label'][lo:lo + n])
        assert np.array_equal(s, d['score'][lo:lo + n])
        lo += n


def _segm_fixture(golden_dir):
    import os
    d = np.load(os.path.join(golden_dir, 'segm_results.npz'))
    D, M = d['logits_sel'].shape[:2]
    logits = np.zeros((D, int(d['n_fg']), M, M), np.float32)
    logits[np.arange(D), d['label']] = d['logits_sel']
    shape = tuple(d['masks_shape'])
    masks = np.unpackbits(d['masks'], axis=-1)[..., :shape[-1]].astype(bool)
    return d['bbox'], d['label'], logits, int(d['im_h']), int(d['im_w']), masks


def test_image_io_pinned_to_reference_bodies(golden_dir):
    """prepare.npz / segm_results.npz come from the reference's own `MaskRCNN.prepare`
    (models/mask_rcnn.py:152-176) and `segm_results` (:63-107) with cv2.resize mapped to the
    oracle's INTER_LINEAR restatement (oracle/gen_golden.py section 9)."""
    import os
    from oracle import np_infer
    d = np.load(os.path.join(golden_dir, 'prepare.npz'))
    for i in range(3):
        out, scale = np_infer.prepare(d['img%d' % i], d['mean'], int(d['min_size']), int(d['max_size']))
        assert scale == float(d['scales'][i])
        assert out.dtype == np.float32 and np.array_equal(out, d['out%d' % i])
    bbox, label, logits, im_h, im_w, masks = _segm_fixture(golden_dir)
    got = np_infer.segm_results(bbox, label, logits, im_h, im_w)
    assert got.shape == masks.shape and np.array_equal(got, masks) and masks.any()
