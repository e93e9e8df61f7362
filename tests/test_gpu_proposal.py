"""Integer path on the device (decode, top-k, NMS, ProposalCreator pieces) — bit-exact
against the oracle on the same inputs."""
import numpy as np
import pytest
import torch

import oracle
from oracle import np_ref
from chainer_mask_rcnn_amd.functions import proposal_ops as P

pytestmark = pytest.mark.gpu


def _rand_boxes(rng, n, size=800.):
    cy, cx = rng.uniform(0, size, n), rng.uniform(0, size, n)
    h, w = rng.uniform(4, 300, n), rng.uniform(4, 300, n)
    b = np.stack([cy - h / 2, cx - w / 2, cy + h / 2, cx + w / 2], 1)
    return np.clip(b, 0, size).astype(np.float32)


@pytest.mark.parametrize('n', [1, 63, 64, 65, 500, 3000])
@pytest.mark.parametrize('thresh', [0.5, 0.7])
def test_nms_sorted_bit_exact(dev, n, thresh):
    rng = np.random.RandomState(n)
    bbox = _rand_boxes(rng, n)
    keep, n_keep = P.nms_sorted(torch.tensor(bbox, device=dev), thresh)
    k = keep[:int(n_keep.item())].cpu().numpy()
    ref = oracle.nms_sorted(bbox, thresh)
    assert k.dtype == np.int32 and np.array_equal(k, ref)


def test_nms_limit_and_dense_overlap(dev):
    rng = np.random.RandomState(7)
    # heavy overlap: jittered copies of a few boxes
    base = _rand_boxes(rng, 20)
    bbox = (base[rng.randint(0, 20, 5000)] + rng.uniform(-3, 3, (5000, 4))).astype(np.float32)
    for limit in (0, 1, 17, 2000):
        keep, n_keep = P.nms_sorted(torch.tensor(bbox, device=dev), 0.7, limit=limit)
        k = keep[:int(n_keep.item())].cpu().numpy()
        ref = oracle.nms_sorted(bbox, 0.7, limit if limit > 0 else -1)
        assert np.array_equal(k, ref)


@pytest.mark.parametrize('n', [511, 512, 513, 575, 1024, 1025, 4097, 12000])
def test_nms_super_step_structure(dev, n):
    """csrc/nms.hip walks the 64-box chunks eight to a super-step (the coupling between the chunks
    of a super-step comes from pre-loaded words, the rest from one bulk phase per super-step):
    box counts around the super-step size, a staircase of boxes in which every box overlaps its
    successors across chunk AND super-step boundaries (long dependency chains), and limits that
    are reached in every position of a super-step."""
    rng = np.random.RandomState(n)
    # staircase: box i = [i * d, i * d + 40] squares; IoU with box i + k falls with k
    d = rng.uniform(1.0, 9.0)
    off = np.arange(n, dtype=np.float64) * d % 700.0
    stair = np.stack([off, off, off + 40.0, off + 40.0], 1).astype(np.float32)
    mixed = np.where((rng.rand(n, 1) < 0.5), stair, _rand_boxes(rng, n)).astype(np.float32)
    for bbox, thresh in ((stair, 0.5), (mixed, 0.7)):
        full = oracle.nms_sorted(bbox, thresh)
        limits = [0] + sorted(set(int(x) for x in (1, len(full) // 3, len(full) // 2 + 1, len(full) - 1, len(full),
                                                   len(full) + 5) if x > 0))
        for limit in limits:
            keep, n_keep = P.nms_sorted(torch.tensor(bbox, device=dev), thresh, limit=limit)
            k = keep[:int(n_keep.item())].cpu().numpy()
            ref = oracle.nms_sorted(bbox, thresh, limit if limit > 0 else -1)
            assert np.array_equal(k, ref), (n, thresh, limit, len(k), len(ref))


def test_nms_degenerate(dev):
    bbox = np.array([[5, 5, 5, 5], [5, 5, 5, 5], [0, 0, 10, 10], [0, 0, 10, 10]], np.float32)
    keep, n_keep = P.nms_sorted(torch.tensor(bbox, device=dev), 0.5)
    assert list(keep[:int(n_keep.item())].cpu().numpy()) == [0, 1, 2]


def test_nms_device_count(dev):
    rng = np.random.RandomState(9)
    bbox = _rand_boxes(rng, 700)
    n_dev = torch.tensor([333], dtype=torch.int32, device=dev)
    keep, n_keep = P.nms_sorted(torch.tensor(bbox, device=dev), 0.6, n_dev=n_dev)
    k = keep[:int(n_keep.item())].cpu().numpy()
    assert np.array_equal(k, oracle.nms_sorted(bbox[:333], 0.6))


def test_nms_batched(dev):
    rng = np.random.RandomState(10)
    G, n_max = 9, 300
    bbox = np.stack([_rand_boxes(rng, n_max) for _ in range(G)])
    counts = rng.randint(0, n_max + 1, G).astype(np.int32)
    counts[0] = 0
    keep, n_keep = P.nms_sorted_batched(torch.tensor(bbox, device=dev),
                                        torch.tensor(counts, device=dev), 0.5)
    keep, n_keep = keep.cpu().numpy(), n_keep.cpu().numpy()
    for g in range(G):
        ref = oracle.nms_sorted(bbox[g, :counts[g]], 0.5)
        assert np.array_equal(keep[g, :n_keep[g]], ref)


def test_non_maximum_suppression_api(dev):
    rng = np.random.RandomState(11)
    bbox = _rand_boxes(rng, 900)
    score = rng.standard_normal(900).astype(np.float32)
    k = P.non_maximum_suppression(torch.tensor(bbox, device=dev), 0.5,
                                  torch.tensor(score, device=dev), limit=50)
    ref = np_ref.non_maximum_suppression(bbox, 0.5, score, 50)
    assert k.dtype == torch.int32 and np.array_equal(k.cpu().numpy(), ref)
    e = P.non_maximum_suppression(torch.zeros((0, 4), device=dev), 0.5)
    assert e.numel() == 0 and e.dtype == torch.int32


@pytest.mark.parametrize('n,k', [(10, 4), (1000, 1000), (64260, 12000), (5000, 6000)])
def test_topk_desc_bit_exact(dev, n, k):
    rng = np.random.RandomState(n)
    score = rng.standard_normal(n).astype(np.float32)
    score[rng.randint(0, n, n // 10)] = score[0]          # ties
    score[rng.randint(0, n, 3)] = 0.0
    score[rng.randint(0, n, 3)] = -0.0
    order, n_out = P.topk_desc(torch.tensor(score, device=dev), k)
    kk = min(k, n)
    ref = np_ref.stable_argsort_desc(score)[:kk]
    assert int(n_out.item()) == kk
    assert np.array_equal(order[:kk].cpu().numpy(), ref.astype(np.int32))


@pytest.mark.parametrize('kind', ['all_equal', 'two_values', 'narrow', 'signed_zero_inf'])
def test_topk_degenerate_score_distributions(dev, kind):
    """The bucketed rank (top 16 key bits) must stay exact when a bucket holds everything
    (all scores equal: pure index order) or when scores straddle sign / zero / infinities."""
    rng = np.random.RandomState(11)
    n, k = 64260, 12000
    if kind == 'all_equal':
        score = np.full(n, 0.25, np.float32)
    elif kind == 'two_values':
        score = rng.choice(np.array([-1.5, 3.0], np.float32), n)
    elif kind == 'narrow':        # one exponent, 7 equal leading mantissa bits: one bucket
        score = (1.0 + rng.randint(0, 1 << 16, n).astype(np.float32) * 2.0 ** -23).astype(np.float32)
    else:
        score = rng.standard_normal(n).astype(np.float32)
        score[rng.randint(0, n, 500)] = 0.0
        score[rng.randint(0, n, 500)] = -0.0
        score[rng.randint(0, n, 50)] = np.inf
        score[rng.randint(0, n, 50)] = -np.inf
    order, n_out = P.topk_desc(torch.tensor(score, device=dev), k)
    ref = np_ref.stable_argsort_desc(score)[:k].astype(np.int32)
    assert int(n_out.item()) == k
    assert np.array_equal(order.cpu().numpy()[:k], ref)


def test_topk_with_validity(dev):
    rng = np.random.RandomState(12)
    n = 3000
    score = rng.standard_normal(n).astype(np.float32)
    valid = (rng.uniform(size=n) > 0.4)
    order, n_out = P.topk_desc(torch.tensor(score, device=dev), 2500,
                               torch.tensor(valid.astype(np.uint8), device=dev))
    idx = np.where(valid)[0]
    ref = idx[np_ref.stable_argsort_desc(score[idx])][:2500]
    assert int(n_out.item()) == len(ref)
    assert np.array_equal(order[:len(ref)].cpu().numpy(), ref.astype(np.int32))


def test_decode_clip_bit_exact(dev):
    rng = np.random.RandomState(13)
    ab = np_ref.generate_anchor_base(16, (0.5, 1, 2), (2, 4, 8, 16, 32))
    anchor = np_ref.enumerate_shifted_anchor(ab, 16, 51, 84)
    loc = (rng.standard_normal((len(anchor), 4)) * 0.5).astype(np.float32)
    roi, valid = P.decode_clip(torch.tensor(anchor, device=dev), torch.tensor(loc, device=dev),
                               (800, 1333), 0.)
    ref = np_ref.loc2bbox(anchor, loc)
    ref[:, 0::2] = np.clip(ref[:, 0::2], 0, 800)
    ref[:, 1::2] = np.clip(ref[:, 1::2], 0, 1333)
    got = roi.cpu().numpy()
    # exp in double then one rounding on both sides: the fp32 boxes are bit-identical
    assert np.array_equal(got, ref)
    assert valid.cpu().numpy().all()


def test_full_proposal_pipeline_matches_oracle(dev):
    """decode -> top-k -> gather -> NMS -> gather at the BASELINE C2 size
    (64260 anchors, 12000 pre-NMS, 2000 post-NMS): same RoIs as the oracle."""
    rng = np.random.RandomState(14)
    ab = np_ref.generate_anchor_base(16, (0.5, 1, 2), (2, 4, 8, 16, 32))
    anchor = np_ref.enumerate_shifted_anchor(ab, 16, 51, 84)
    loc = (rng.standard_normal((len(anchor), 4)) * 0.2).astype(np.float32)
    score = rng.standard_normal(len(anchor)).astype(np.float32)
    pc = np_ref.ProposalCreator(min_size=0, n_test_pre_nms=6000, n_test_post_nms=1000)
    ref_roi, ref_idx = pc(loc, score, anchor, (800, 1333), 1.6, train=True, return_indices=True)

    a, l, s = (torch.tensor(v, device=dev) for v in (anchor, loc, score))
    roi, valid = P.decode_clip(a, l, (800, 1333), 0.)
    order, n_sorted = P.topk_desc(s, 12000, valid)
    sroi = P.gather_rows(roi, order, n_sorted)
    keep, n_keep = P.nms_sorted(sroi, 0.7, n_sorted, limit=2000)
    nk = int(n_keep.item())
    out = P.gather_rows(sroi, keep[:nk].contiguous())
    idx = order[keep[:nk].long()].cpu().numpy()
    assert nk == len(ref_roi)
    assert np.array_equal(idx, ref_idx)
    assert np.array_equal(out.cpu().numpy(), ref_roi)


def test_topk_batched_equals_per_row(dev):
    rng = np.random.RandomState(5)
    G, n, k = 3, 20000, 6000
    score = rng.standard_normal((G, n)).astype(np.float32)
    score[1, rng.randint(0, n, 2000)] = 0.5
    valid = (rng.uniform(size=(G, n)) > 0.3).astype(np.uint8)
    valid[2] = 1
    order, n_out = P.topk_desc_batched(torch.tensor(score, device=dev), k, torch.tensor(valid, device=dev))
    order, n_out = order.cpu().numpy(), n_out.cpu().numpy()
    for g in range(G):
        idx = np.where(valid[g] > 0)[0]
        ref = idx[np_ref.stable_argsort_desc(score[g][idx])][:k]
        assert n_out[g] == len(ref)
        assert np.array_equal(order[g, :len(ref)], ref.astype(np.int32))
