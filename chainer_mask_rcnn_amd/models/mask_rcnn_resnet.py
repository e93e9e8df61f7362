"""MaskRCNNResNet and its RoI head — same constructor, attributes and call signatures as
the reference's /root/reference/chainer_mask_rcnn/models/mask_rcnn_resnet.py:30-196.

Head (ResNetRoIHead.__call__, :168-196): ROIAlign 14x14 @1/16 (axes='yx') -> res5
(stride roi_size//7) -> {avg-pool 7 -> cls_loc, score} and {deconv6 2x2/2 + ReLU -> mask 1x1}.
``cls_loc`` and ``score`` read the same pooled vector, so they run as one GEMM over a fused
(4*n_class + n_class, padded to a multiple of 4) weight; ``cls_loc.W`` / ``score.W`` are views.
"""
import numpy as np
import torch

from .. import functions
from .. import functions as F
from .mask_rcnn import MaskRCNN
from .region_proposal_network import RegionProposalNetwork, _ConvView
from .resnet_extractor import BuildingBlock, Convolution2D
from .resnet_extractor import ResNet101Extractor
from .resnet_extractor import ResNet50Extractor


class _Linear(torch.nn.Module):
    """L.Linear parameter holder: W (out, in), b (out,)."""

    def __init__(self, in_size, out_size):
        super(_Linear, self).__init__()
        self.W = torch.nn.Parameter(torch.zeros((out_size, in_size), dtype=torch.float32))
        self.b = torch.nn.Parameter(torch.zeros((out_size,), dtype=torch.float32))


class _Deconvolution2D(torch.nn.Module):
    """L.Deconvolution2D(in, out, 2, stride=2): W (in, out, 2, 2) stored channels-last
    (= (in, 2, 2, out), the KRSC filter of the adjoint 2x2/2 convolution), b (out,)."""

    def __init__(self, in_ch, out_ch, std):
        super(_Deconvolution2D, self).__init__()
        w = torch.empty((in_ch, 2, 2, out_ch), dtype=torch.float32).permute(0, 3, 1, 2)
        self.W = torch.nn.Parameter(w)
        with torch.no_grad():
            self.W.normal_(0., std)
        self.b = torch.nn.Parameter(torch.zeros((out_ch,), dtype=torch.float32))


class ResNetRoIHead(torch.nn.Module):

    mask_size = 14  # Size of the predicted mask.

    def __init__(self, n_layers, n_class, roi_size, spatial_scale, pretrained_model=None,
                 res_initialW=None, loc_initialW=None, score_initialW=None,
                 mask_initialW=None, pooling_func=functions.roi_align_2d):
        super(ResNetRoIHead, self).__init__()
        # n_class includes the background
        self.res5 = BuildingBlock(3, 1024, 512, 2048, stride=roi_size // 7)
        n_fc = ((5 * n_class + 3) // 4) * 4
        self.cls_loc_score = _Linear(2048, n_fc)
        loc_std = 0.001 if loc_initialW is None else float(loc_initialW)
        score_std = 0.01 if score_initialW is None else float(score_initialW)
        mask_std = 0.01 if mask_initialW is None else float(mask_initialW)
        with torch.no_grad():
            self.cls_loc_score.W[:4 * n_class].normal_(0., loc_std)
            self.cls_loc_score.W[4 * n_class:5 * n_class].normal_(0., score_std)
        self.cls_loc = _ConvView(self.cls_loc_score, 0, 4 * n_class)
        self.score = _ConvView(self.cls_loc_score, 4 * n_class, 5 * n_class)
        # 7 x 7 x 2048 -> 14 x 14 x 256 -> 14 x 14 x n_fg_class
        self.deconv6 = _Deconvolution2D(2048, 256, mask_std)
        self.mask = Convolution2D(256, n_class - 1, 1, std=mask_std)

        self.n_class = n_class
        self.roi_size = roi_size
        self.spatial_scale = spatial_scale
        self.pooling_func = pooling_func
        self.fused_tail = True      # developer switch: res5's two consumers inside the stage node
        # Pool BEHIND res5.a's 1x1 projections instead of in front of them (functions/conv.py
        # "projected pooling": the same values up to fp32 rounding, the two projections on the map's
        # pixels instead of the pooled ones).  None = follow functions.conv.PROJECTED_POOLING.
        self.projected_pooling = None

    def forward(self, x, rois, roi_indices, pred_bbox=True, pred_mask=True, mask_rows=None):
        """Reference: models/mask_rcnn_resnet.py:168-196.

        ``mask_rows`` (extension): int64 index tensor of the RoI rows the mask branch is run
        for; ``roi_masks`` then has ``len(mask_rows)`` rows.  ``None`` = all rows."""
        from .. import optimizers
        optimizers.join_pending_all()      # deferred updates of the head's parameters (if any)
        if rois.shape[0] == 0:
            # nothing to pool (an image without proposals): empty outputs of the right widths
            z = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=x.device)
            return (z(0, 4 * self.n_class) if pred_bbox else None, z(0, self.n_class) if pred_bbox else None,
                    z(0, self.n_class - 1, self.mask_size, self.mask_size) if pred_mask else None)
        # (batch, x1, y1, x2, y2) rows when the caller built them with the RoIs (MaskRCNNTrainChain)
        rois5 = getattr(rois, '_mrcnn_rois5', None)
        if rois5 is not None and tuple(rois5.shape) != (rois.shape[0], 5):
            rois5 = None
        indices_and_rois = None
        if rois5 is None:
            roi_indices = roi_indices.to(torch.float32)
            indices_and_rois = torch.cat((roi_indices[:, None], rois), dim=1)
        res5_stride = self.roi_size // 7
        # Both consumers of res5 wanted and the mask branch on a row subset (training): the fused
        # stage hands back (pool5, res5[mask_rows]) and combines their gradients in one pass
        fused_tail = pred_bbox and pred_mask and mask_rows is not None and \
            getattr(self.res5, 'fused_stage', False) and self.fused_tail
        kw = dict(tail_rows=mask_rows) if fused_tail else {}
        from ..functions import conv as _conv
        projected = self.projected_pooling if self.projected_pooling is not None else _conv.PROJECTED_POOLING
        projected = projected and self.pooling_func is functions.roi_align_2d and \
            getattr(self.res5, 'fused_stage', False) and rois.shape[0] > 0
        if not projected and indices_and_rois is None:
            indices_and_rois = rois5[:, [0, 2, 1, 4, 3]]      # the reference-order branches take 'yx' rows
        if projected:
            # the stage node pools inside block a: conv1 / conv4 (1x1, stride s) read only the bins
            # (s*i, s*j) of the pooled map, and they commute with ROIAlign
            order = getattr(rois, '_mrcnn_order', None)
            # without a graph (predict: one head call per image and per mask group on ONE map) the
            # two projections of the map are computed once and kept until the map or the weights change
            pre = None
            if not (torch.is_grad_enabled() and (x.requires_grad or self.res5.a.conv1.W.requires_grad)):
                pre = _conv.projected_map(x, self.res5.a.conv1.W, self.res5.a.conv4.W)
            spec = _conv.RoiSpec(rois5 if rois5 is not None else indices_and_rois[:, [0, 2, 1, 4, 3]],
                                 self.roi_size, self.roi_size,
                                 self.spatial_scale, bin_stride=res5_stride, order=order, proj=pre)
            res5 = self.res5(x, first_stride=1, roi=spec, **kw)
        elif res5_stride > 1 and self.pooling_func is functions.roi_align_2d:
            # res5.a reads the pooled map only through 1x1 stride-s convolutions (conv1 and
            # the shortcut conv4), i.e. only the bins (s*i, s*j): pool just those and run the
            # block with stride 1 — same values, a quarter of the ROIAlign work and traffic.
            # a spatially sorted processing order, when the caller sampled the RoIs on the host
            # and attached one (MaskRCNNTrainChain): same values, fewer feature-map re-reads
            order = getattr(rois, '_mrcnn_order', None)
            okw = dict(order=order) if order is not None else {}
            pool = self.pooling_func(
                x, indices_and_rois, outh=self.roi_size, outw=self.roi_size,
                spatial_scale=self.spatial_scale, axes='yx', bin_stride=res5_stride, **okw)
            res5 = self.res5(pool, first_stride=1, **kw)
        else:
            pool = self.pooling_func(
                x, indices_and_rois, outh=self.roi_size, outw=self.roi_size,
                spatial_scale=self.spatial_scale, axes='yx')
            res5 = self.res5(pool, **kw)

        roi_cls_locs = roi_scores = roi_masks = None
        res5_fg = None
        pool5 = None
        if fused_tail:
            pool5, res5_fg = res5
        elif pred_bbox and pred_mask and mask_rows is not None:
            res5, res5_fg = F.fanout_rows(res5, mask_rows)
        if pred_bbox:
            if pool5 is None:
                pool5 = F.average_pooling_2d(res5, 7, stride=7)
            fc = F.linear(pool5, self.cls_loc_score.W, self.cls_loc_score.b)
            roi_cls_locs = fc[:, :4 * self.n_class]
            roi_scores = fc[:, 4 * self.n_class:5 * self.n_class]
        if pred_mask:
            if res5_fg is not None:
                res5 = res5_fg
            elif mask_rows is not None:
                res5 = res5.index_select(0, mask_rows)
            deconv6 = F.deconv2x2s2(res5, self.deconv6.W, self.deconv6.b, relu=True)
            roi_masks = self.mask(deconv6)
        return roi_cls_locs, roi_scores, roi_masks


class MaskRCNNResNet(MaskRCNN):

    feat_stride = 16

    def __init__(self, n_layers, n_fg_class, pretrained_model=None, min_size=600,
                 max_size=1000, ratios=(0.5, 1, 2), anchor_scales=(4, 8, 16, 32),
                 mean=(123.152, 115.903, 103.063), res_initialW=None, rpn_initialW=None,
                 loc_initialW=None, score_initialW=None, mask_initialW=None,
                 proposal_creator_params=dict(min_size=0, n_test_pre_nms=6000,
                                              n_test_post_nms=1000),
                 pooling_func=functions.roi_align_2d, rpn_hidden=1024, roi_size=7):
        if n_layers == 50:
            extractor = ResNet50Extractor(remove_layers=['res5', 'fc6'])
        elif n_layers == 101:
            extractor = ResNet101Extractor(remove_layers=['res5', 'fc6'])
        else:
            raise ValueError

        rpn = RegionProposalNetwork(
            1024, rpn_hidden, ratios=ratios, anchor_scales=anchor_scales,
            feat_stride=self.feat_stride, initialW=rpn_initialW,
            proposal_creator_params=proposal_creator_params)
        head = ResNetRoIHead(
            n_layers=n_layers, n_class=n_fg_class + 1, roi_size=roi_size,
            spatial_scale=1. / self.feat_stride, res_initialW=res_initialW,
            loc_initialW=loc_initialW, score_initialW=score_initialW,
            mask_initialW=mask_initialW, pooling_func=pooling_func)

        if len(mean) != 3:
            raise ValueError('The mean must be tuple of RGB values.')
        mean = np.asarray(mean, dtype=np.float32)[:, None, None]

        super(MaskRCNNResNet, self).__init__(
            extractor, rpn, head, mean=mean, min_size=min_size, max_size=max_size)

        if pretrained_model:
            from ..serializers import load_npz
            load_npz(pretrained_model, self)
