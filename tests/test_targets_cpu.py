"""Host-side target creators of the product (NumPy, like the reference) vs the oracle
restatements with the same global RNG seed: integer outputs must be identical."""
import numpy as np
import pytest

from oracle import np_ref, np_targets
from chainer_mask_rcnn_amd.models.utils import ProposalTargetCreator, AnchorTargetCreator
from chainer_mask_rcnn_amd.utils import bbox as B


def _scene(seed, H=480, W=640, G=5, R=600):
    rng = np.random.RandomState(seed)
    y0 = rng.uniform(0, H - 80, G); x0 = rng.uniform(0, W - 80, G)
    bbox = np.stack([y0, x0, y0 + rng.uniform(40, 200, G), x0 + rng.uniform(40, 200, G)], 1)
    bbox[:, 2] = np.minimum(bbox[:, 2], H); bbox[:, 3] = np.minimum(bbox[:, 3], W)
    bbox = bbox.astype(np.float32)
    label = rng.randint(0, 80, G).astype(np.int32)
    yy, xx = np.mgrid[0:H, 0:W]
    mask = np.zeros((G, H, W), np.int32)
    for g in range(G):
        cy, cx = (bbox[g, 0] + bbox[g, 2]) / 2, (bbox[g, 1] + bbox[g, 3]) / 2
        ry, rx = (bbox[g, 2] - bbox[g, 0]) / 2, (bbox[g, 3] - bbox[g, 1]) / 2
        mask[g] = (((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0)
    # proposals: jittered ground truth + random boxes
    jit = bbox[rng.randint(0, G, R // 2)] + rng.uniform(-25, 25, (R // 2, 4))
    ry0 = rng.uniform(0, H - 20, R - R // 2); rx0 = rng.uniform(0, W - 20, R - R // 2)
    rnd = np.stack([ry0, rx0, ry0 + rng.uniform(10, 300, len(ry0)), rx0 + rng.uniform(10, 300, len(ry0))], 1)
    roi = np.concatenate([jit, rnd], 0)
    roi[:, 0::2] = np.clip(roi[:, 0::2], 0, H); roi[:, 1::2] = np.clip(roi[:, 1::2], 0, W)
    return roi.astype(np.float32), bbox, label, mask, (H, W)


@pytest.mark.parametrize('seed', [0, 1, 2])
def test_proposal_target_creator_matches_oracle(seed):
    roi, bbox, label, mask, _ = _scene(seed)
    np.random.seed(7)
    got = ProposalTargetCreator(n_sample=128)(roi, bbox, label, mask)
    np.random.seed(7)
    ref = np_targets.ProposalTargetCreator(n_sample=128)(roi, bbox, label, mask)
    assert np.array_equal(got[0], ref[0])                       # sampled RoIs
    np.testing.assert_allclose(got[1], ref[1], rtol=1e-6, atol=1e-6)
    assert got[2].dtype == np.int32 and np.array_equal(got[2], ref[2])
    assert got[3].dtype == np.int32 and np.array_equal(got[3], ref[3])
    n_fg = int((got[2] > 0).sum())
    assert 0 < n_fg <= 32 and (got[3][n_fg:] == -1).all() and set(np.unique(got[3][:n_fg])) <= {0, 1}


def test_proposal_target_creator_errors_and_counts():
    roi, bbox, label, mask, _ = _scene(3)
    with pytest.raises(ValueError):
        ProposalTargetCreator()(roi, np.zeros((0, 4), np.float32), label[:0], mask[:0])
    np.random.seed(0)
    s_roi, loc, lab, m = ProposalTargetCreator()(roi, bbox, label, mask)
    assert len(s_roi) == len(loc) == len(lab) == len(m) <= 512
    assert (lab > 0).sum() <= 128


@pytest.mark.parametrize('seed', [0, 1])
def test_anchor_target_creator_matches_oracle(seed):
    _, bbox, _, _, size = _scene(seed)
    ab = np_ref.generate_anchor_base(16, (0.5, 1, 2), (2, 4, 8, 16, 32))
    anchor = np_ref.enumerate_shifted_anchor(ab, 16, 30, 40)
    np.random.seed(11)
    loc, label = AnchorTargetCreator()(bbox, anchor, size)
    np.random.seed(11)
    loc_r, label_r = np_ref.AnchorTargetCreator()(bbox, anchor, size)
    assert np.array_equal(label, label_r)
    assert np.array_equal(loc, loc_r)


def test_host_bbox_utils_match_oracle():
    rng = np.random.RandomState(0)
    a = rng.uniform(0, 400, (300, 4)).astype(np.float32); a[:, 2:] += a[:, :2]
    b = rng.uniform(0, 400, (7, 4)).astype(np.float32); b[:, 2:] += b[:, :2]
    assert np.array_equal(B.bbox_iou(a, b), np_ref.bbox_iou(a, b))
    assert np.array_equal(B.bbox2loc(a[:7], b), np_ref.bbox2loc(a[:7], b))
    ab = B.generate_anchor_base(16, (0.5, 1, 2), (2, 4, 8, 16, 32))
    assert np.array_equal(ab, np_ref.generate_anchor_base(16, (0.5, 1, 2), (2, 4, 8, 16, 32)))
    assert np.array_equal(B.enumerate_shifted_anchor(ab, 16, 51, 84),
                          np_ref.enumerate_shifted_anchor(ab, 16, 51, 84))
    img = rng.uniform(size=(23, 31)).astype(np.float32)
    assert np.array_equal(B.resize_bilinear(img, 14, 14), np_ref.resize_bilinear(img, 14, 14))


def test_proposal_target_creator_pinned_to_reference_class(golden_dir):
    """Fixture produced by the reference's own ProposalTargetCreator class body
    (models/utils/proposal_target_creator.py:25-184) running on the oracle's restatements of
    bbox_iou / bbox2loc / cv2.resize (oracle/gen_golden.py section 6): sampled RoIs, labels,
    14x14 mask targets and the position of the global np.random stream afterwards must be
    identical for the oracle's literal restatement and for the product's split sample /
    mask_targets implementation."""
    import os
    d = np.load(os.path.join(golden_dir, 'proposal_target_creator.npz'))
    mask = d['mask'].astype(np.int32)
    for make in (np_targets.ProposalTargetCreator, ProposalTargetCreator):
        np.random.seed(int(d['seed']))
        s_roi, loc, lab, m = make(n_sample=int(d['n_sample']))(d['roi'], d['bbox'], d['label'], mask)
        assert np.random.randint(0, 1 << 30) == int(d['next_randint'])
        assert np.array_equal(s_roi, d['sample_roi'])
        assert lab.dtype == np.int32 and np.array_equal(lab, d['gt_roi_label'])
        np.testing.assert_allclose(loc, d['gt_roi_loc'], rtol=1e-6, atol=1e-6)
        assert m.dtype == np.int32 and np.array_equal(m, d['gt_roi_mask'])
    # the product's two halves, as MaskRCNNTrainChain calls them
    ptc = ProposalTargetCreator(n_sample=int(d['n_sample']))
    np.random.seed(int(d['seed']))
    s_roi, loc, lab, job = ptc.sample(d['roi'], d['bbox'], d['label'])
    assert np.random.randint(0, 1 << 30) == int(d['next_randint'])   # mask_targets draws nothing
    assert np.array_equal(ptc.mask_targets(job, mask), d['gt_roi_mask'])


def test_degenerate_crop_is_all_background():
    """Documented deviation (proposal_target_creator.mask_targets): a foreground RoI whose
    rounded box is empty makes the reference raise on the empty crop; the split job returns an
    all-background (0) mask target for it and -1 rows for the background RoIs."""
    H, W = 64, 64
    mask = np.ones((1, H, W), np.int32)
    ptc = ProposalTargetCreator(n_sample=4)
    boxes = np.array([[10, 10, 10, 30], [5, 5, 25, 25]], np.int32)   # first: zero height
    out = ptc.mask_targets((4, 2, boxes, np.array([0, 0])), mask)
    assert out.shape == (4, 14, 14) and out.dtype == np.int32
    assert (out[0] == 0).all() and (out[1] == 1).all() and (out[2:] == -1).all()
