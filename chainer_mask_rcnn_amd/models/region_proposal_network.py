"""Region Proposal Network — same interface as the reference's
/root/reference/chainer_mask_rcnn/models/region_proposal_network.py:26-167.

``conv1`` 3x3 (+bias, ReLU fused) then the two 1x1 heads.  The ``loc`` (4A) and
``score`` (A) convolutions share their input, so they run as ONE implicit-GEMM launch
over a fused (4A + A, padded to a multiple of 4) filter; ``loc.W`` / ``score.W`` are
views of that filter with the reference's shapes.
"""
import numpy as np
import torch

from .. import functions as F
from ..utils.bbox import generate_anchor_base, enumerate_shifted_anchor
from .resnet_extractor import Convolution2D
from .utils.proposal_creator import ProposalCreator


class _ConvView(object):
    """Read-only view of rows [lo, hi) of a fused 1x1 filter / bias."""

    def __init__(self, owner, lo, hi):
        self._owner, self._lo, self._hi = owner, lo, hi

    @property
    def W(self):
        return self._owner.W[self._lo:self._hi]

    @property
    def b(self):
        return self._owner.b[self._lo:self._hi]


class RegionProposalNetwork(torch.nn.Module):

    def __init__(self, in_channels=512, mid_channels=512, ratios=[0.5, 1, 2],
                 anchor_scales=[8, 16, 32], feat_stride=16, initialW=None,
                 proposal_creator_params=dict()):
        super(RegionProposalNetwork, self).__init__()
        self.anchor_base = generate_anchor_base(anchor_scales=anchor_scales, ratios=ratios)
        self.feat_stride = feat_stride
        self.proposal_layer = ProposalCreator(**proposal_creator_params)
        std = 0.01 if initialW is None else float(initialW)
        A = self.anchor_base.shape[0]
        self.n_anchor = A
        self.conv1 = Convolution2D(in_channels, mid_channels, 3, 1, 1, std=std)
        n_out = ((5 * A + 3) // 4) * 4
        self.loc_score = Convolution2D(mid_channels, n_out, 1, 1, 0, std=std)
        with torch.no_grad():
            self.loc_score.W[5 * A:].zero_()
        self.loc = _ConvView(self.loc_score, 0, 4 * A)
        self.score = _ConvView(self.loc_score, 4 * A, 5 * A)
        self._anchor_cache = {}
        # map positions outside which the gradient of conv1's output is exactly zero (the sampled
        # anchors of the RPN losses): filled by MaskRCNNTrainChain where the anchor targets are
        # built, consumed by conv1's backward (functions/conv.py: SparseRows).  A NEW object per
        # forward: each graph's conv1 node keeps the hint of ITS forward, so two forwards before a
        # backward (gradient accumulation) never see each other's rows
        self.grad_rows = None

    def _anchor(self, hh, ww, device):
        key = (hh, ww, str(device))
        if key not in self._anchor_cache:
            a = enumerate_shifted_anchor(self.anchor_base, self.feat_stride, hh, ww)
            self._anchor_cache[key] = (a, torch.tensor(a, device=device))
        return self._anchor_cache[key]

    def forward(self, x, img_size, scales):
        """x (N,C,H,W) -> rpn_locs (N,HWA,4), rpn_scores (N,HWA), rois (R',4),
        roi_indices (R',) int32, anchor (HWA,4)."""
        n, _, hh, ww = x.shape
        A = self.n_anchor
        _, anchor = self._anchor(hh, ww, x.device)
        from ..functions.conv import SparseRows, sparse_output_grad
        self.grad_rows = SparseRows()
        with sparse_output_grad(self.grad_rows):
            h = self.conv1(x, relu=True)
        out = self.loc_score(h)                               # (N, 5A(+pad), H, W), NHWC
        nhwc = out.permute(0, 2, 3, 1)
        rpn_locs = nhwc[..., :4 * A].reshape(n, -1, 4)
        rpn_scores = nhwc[..., 4 * A:5 * A].reshape(n, -1)

        # the proposal window opens (top-k / NMS + the host's RoI sampling: the GPU is nearly
        # idle): weight gradients the optimizer held back from the previous step run here
        from .. import optimizers
        if optimizers.DEFER_LAUNCH_AT == 'window-open':
            optimizers.launch_pending_all()
        self.proposal_layer.train = self.training
        if hasattr(self.proposal_layer, 'batch'):
            rois = self.proposal_layer.batch(rpn_locs, rpn_scores, anchor, img_size, scales)
        else:
            rois = [self.proposal_layer(rpn_locs[i], rpn_scores[i], anchor, img_size,
                                        scale=float(scales[i])) for i in range(n)]
        if optimizers.DEFER_LAUNCH_AT != 'window-open':
            optimizers.launch_pending_all()    # the proposal read-back has returned: GPU idle
        self.last_counts = [int(len(roi)) for roi in rois]     # host-known: no read-back
        roi_indices = [torch.full((len(roi),), i, dtype=torch.int32, device=x.device)
                       for i, roi in enumerate(rois)]
        rois = torch.cat(rois, dim=0)
        roi_indices = torch.cat(roi_indices, dim=0)
        return rpn_locs, rpn_scores, rois, roi_indices, anchor

    def host_anchor(self, hh, ww, device):
        return self._anchor(hh, ww, device)[0]
