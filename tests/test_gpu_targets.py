"""Device half of the target creators vs the host creators — which
tests/test_targets_cpu.py pins to the oracle and to the reference's own class body — with the
same global np.random seed: sampled sets, labels and mask targets are integer results and must
be identical; regression targets are fp32 (logs) within 1e-6."""
import numpy as np
import pytest
import torch

import chainer_mask_rcnn_amd as cmr
from chainer_mask_rcnn_amd.functions import target_ops as T
from chainer_mask_rcnn_amd.models.utils import ProposalTargetCreator, AnchorTargetCreator
from chainer_mask_rcnn_amd.utils import bbox as B
from oracle import np_ref
from test_targets_cpu import _scene

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('seed', [0, 1, 2])
def test_proposal_target_creator_device_matches_host(dev, seed):
    roi, bbox, label, mask, _ = _scene(seed)
    ptc = ProposalTargetCreator(n_sample=128)
    np.random.seed(7)
    ref = ptc(roi, bbox, label, mask)
    after_ref = np.random.randint(0, 2 ** 31 - 1)
    np.random.seed(7)
    s_roi, loc, lab, job = ptc.sample_device(torch.tensor(roi, device=dev), bbox, label)
    after = np.random.randint(0, 2 ** 31 - 1)
    assert after == after_ref                                   # same draws, same stream position
    assert np.array_equal(s_roi.cpu().numpy(), ref[0])
    assert lab.dtype == torch.int32 and np.array_equal(lab.cpu().numpy(), ref[2])
    np.testing.assert_allclose(loc.cpu().numpy(), ref[1], rtol=1e-6, atol=1e-6)
    for m in (torch.tensor(mask, device=dev), torch.tensor(mask != 0, device=dev).to(torch.uint8), mask):
        got = ptc.mask_targets_device(job, m)
        assert got.dtype == torch.int32 and np.array_equal(got.cpu().numpy(), ref[3])
    assert job['n_fg'] == int((ref[2] > 0).sum()) > 0


def test_bbox_iou_argmax_matches_oracle(dev):
    rng = np.random.RandomState(3)
    roi, bbox, _, _, _ = _scene(4, R=3000, G=9)
    roi[5] = roi[6]                                             # exact ties in a row
    bbox[3] = bbox[2]                                           # duplicated ground truth: argmax tie
    best, arg, iou, col = T.bbox_iou_argmax(torch.tensor(roi, device=dev),
                                            torch.tensor(bbox, device=dev), want_matrix=True)
    ref = np_ref.bbox_iou(roi, bbox)
    assert np.array_equal(iou.cpu().numpy(), ref)               # fp32, same operation order
    assert np.array_equal(arg.cpu().numpy(), ref.argmax(1))
    assert np.array_equal(best.cpu().numpy(), ref.max(1))
    assert np.array_equal(col.cpu().numpy(), ref.max(0))


@pytest.mark.parametrize('seed', [0, 1])
def test_anchor_target_creator_device_matches_host(dev, seed):
    _, bbox, _, _, size = _scene(seed)
    ab = B.generate_anchor_base(16, (0.5, 1, 2), (2, 4, 8, 16, 32))
    anchor = B.enumerate_shifted_anchor(ab, 16, size[0] // 16, size[1] // 16)
    atc = AnchorTargetCreator()
    np.random.seed(9)
    loc_ref, label_ref = atc(bbox, anchor, size)
    after_ref = np.random.randint(0, 2 ** 31 - 1)
    np.random.seed(9)
    st = atc.prepare_device(bbox, torch.tensor(anchor, device=dev), anchor, size)
    loc, label = atc.finish_device(st)
    assert np.random.randint(0, 2 ** 31 - 1) == after_ref
    assert np.array_equal(label.cpu().numpy(), label_ref)
    assert (label_ref == 1).sum() > 0 and (label_ref == 0).sum() > 0
    np.testing.assert_allclose(loc.cpu().numpy(), loc_ref, rtol=1e-6, atol=1e-6)


def test_train_chain_device_targets_same_step(dev):
    """MaskRCNNTrainChain.device_targets: same sampled RoIs, labels, mask / RPN targets and the
    same six losses as the host creators (host masks AND device-resident masks)."""
    from test_gpu_model import _build
    model, chain, imgs, bboxes, labels, masks = _build(dev)
    x = torch.tensor(imgs, device=dev)
    out = {}
    for mode in ('host', 'device', 'device-masks'):
        chain.device_targets = mode != 'host'
        mm = masks if mode != 'device-masks' else [torch.tensor(m, device=dev) for m in masks]
        np.random.seed(123)
        with torch.no_grad():
            chain(x, bboxes, labels, mm, [1., 1.])
        t = chain.last_targets
        out[mode] = ({k: t[k].cpu().numpy() for k in ('sample_rois', 'gt_roi_labels', 'gt_roi_masks',
                                                       'gt_rpn_labels')},
                     {k: float(v) for k, v in chain.report.items()}, np.random.randint(0, 2 ** 31 - 1))
    chain.device_targets = False
    for mode in ('device', 'device-masks'):
        for k, v in out['host'][0].items():
            assert np.array_equal(out[mode][0][k], v), (mode, k)
        assert out[mode][2] == out['host'][2]
        for k, v in out['host'][1].items():
            assert abs(out[mode][1][k] - v) <= 1e-6 * max(abs(v), 1e-3), (mode, k)
