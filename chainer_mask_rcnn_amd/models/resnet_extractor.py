"""ResNet-50/101 C4 extractor — same structure, parameter names and semantics as the
reference's /root/reference/chainer_mask_rcnn/models/resnet_extractor.py:47-124 built on
chainer's ResNet50Layers / BuildingBlock / BottleneckA/B:

  conv1 7x7/2 (+bias) -> bn1 (AffineChannel2D) -> relu -> max-pool 3x3/2 pad 1 (cover_all)
  -> res2 (3 blocks) [no gradient below: unchain_backward at freeze_at='res2']
  -> res3 (4 blocks, stride 2 in the first 1x1) -> res4 (6 | 23 blocks) = target layer.

Every BatchNormalization is an AffineChannel2D (:32-44) and is fused into the epilogue of
the convolution it follows, together with the residual add and ReLU, so one bottleneck is
three (or four) launches of the implicit-GEMM kernel and nothing else.
"""
import collections
import math

import torch

from .. import functions as F
from ..links import AffineChannel2D


def _he_normal_(w, fan_in):
    with torch.no_grad():
        w.normal_(0., math.sqrt(2. / fan_in))


class Convolution2D(torch.nn.Module):
    """Parameter holder mirroring ``L.Convolution2D`` (W (out,in,kh,kw), optional b),
    stored channels-last = KRSC."""

    def __init__(self, in_ch, out_ch, ksize, stride=1, pad=0, nobias=False, std=None):
        super(Convolution2D, self).__init__()
        w = torch.empty((out_ch, ksize, ksize, in_ch), dtype=torch.float32).permute(0, 3, 1, 2)
        self.W = torch.nn.Parameter(w)
        if std is None:
            _he_normal_(self.W, in_ch * ksize * ksize)
        else:
            with torch.no_grad():
                self.W.normal_(0., std)
        self.b = None if nobias else torch.nn.Parameter(torch.zeros(out_ch))
        self.stride, self.pad = stride, pad

    def forward(self, x, affine=None, residual=None, relu=False):
        scale = shift = None
        if affine is not None:
            scale, shift = affine.W, affine.b
        return F.conv2d(x, self.W, self.b, self.stride, self.pad, scale=scale, shift=shift,
                        residual=residual, relu=relu)


class Bottleneck(torch.nn.Module):
    """chainer BottleneckA (with projection shortcut conv4/bn4) or BottleneckB."""

    def __init__(self, in_ch, mid_ch, out_ch, stride=1, projection=False):
        super(Bottleneck, self).__init__()
        self.conv1 = Convolution2D(in_ch, mid_ch, 1, stride, 0, nobias=True)
        self.bn1 = AffineChannel2D(mid_ch)
        self.conv2 = Convolution2D(mid_ch, mid_ch, 3, 1, 1, nobias=True)
        self.bn2 = AffineChannel2D(mid_ch)
        self.conv3 = Convolution2D(mid_ch, out_ch, 1, 1, 0, nobias=True)
        self.bn3 = AffineChannel2D(out_ch)
        self.projection = projection
        if projection:
            self.conv4 = Convolution2D(in_ch, out_ch, 1, stride, 0, nobias=True)
            self.bn4 = AffineChannel2D(out_ch)

    def forward(self, x, stride=None):
        if self.projection:
            return F.bottleneck(x, self.conv1, self.bn1, self.conv2, self.bn2, self.conv3,
                                self.bn3, self.conv4, self.bn4,
                                stride=self.conv1.stride if stride is None else stride)
        return F.bottleneck(x, self.conv1, self.bn1, self.conv2, self.bn2, self.conv3, self.bn3)

    def forward_unfused(self, x):
        """Same block as four separate fused-conv autograd nodes (kept for testing)."""
        h = self.conv1(x, self.bn1, relu=True)
        h = self.conv2(h, self.bn2, relu=True)
        shortcut = self.conv4(x, self.bn4) if self.projection else x
        return self.conv3(h, self.bn3, residual=shortcut, relu=True)


class BuildingBlock(torch.nn.Module):
    """chainer BuildingBlock(n_layer, in, mid, out, stride): children a, b1 .. b{n-1}.

    ``fused_stage`` (default): the whole stage is one autograd node whose backward GEMMs
    need no mask staging (functions/conv.py:_StageFn); False chains per-bottleneck nodes.
    """
    fused_stage = True

    def __init__(self, n_layer, in_ch, mid_ch, out_ch, stride):
        super(BuildingBlock, self).__init__()
        self.a = Bottleneck(in_ch, mid_ch, out_ch, stride, projection=True)
        self._names = ['a']
        for i in range(n_layer - 1):
            name = 'b{}'.format(i + 1)
            setattr(self, name, Bottleneck(out_ch, mid_ch, out_ch))
            self._names.append(name)
        # optional callable polled during the fused stage's backward (entry + after every
        # block): parallel.DataParallelGradSync launches completed gradient buckets from it
        self.grad_poll = None

    def forward(self, x, first_stride=None, tail_rows=None, roi=None):
        """``first_stride`` overrides the stride of block ``a`` (used by the RoI head when the
        stride-2 subsampling has already been done by the pooling op).  ``tail_rows`` (fused
        stage only): return ``(average_pooling_2d(y), y[tail_rows])`` instead of y.  ``roi`` (fused
        stage only; a ``functions.conv.RoiSpec``): ``x`` is the feature map and the stage pools
        inside block ``a``, behind its 1x1 projections (functions/conv.py "projected pooling")."""
        if self.fused_stage:
            return F.building_block(x, [getattr(self, n) for n in self._names], first_stride,
                                    poll=self.grad_poll, tail_rows=tail_rows, roi=roi)
        if tail_rows is not None or roi is not None:
            raise ValueError('tail_rows / roi need the fused stage')
        for name in self._names:
            if name == 'a' and first_stride is not None:
                x = self.a(x, stride=first_stride)
            else:
                x = getattr(self, name)(x)
        return x


def pad_image_nhwc4(x):
    """(N,3,H,W) logical NCHW image -> dense (N,H,W,4) with a zero 4th channel."""
    n, c, h, w = x.shape
    assert c == 3
    out = torch.zeros((n, h, w, 4), dtype=torch.float32, device=x.device)
    out[..., :3] = x.permute(0, 2, 3, 1)
    return out


def pack_stem_filter(W):
    """conv1 filter (K,3,7,7) -> (K,7,8,4) with zeros at s=7 and c=3 (stem kernel layout)."""
    k = W.shape[0]
    out = torch.zeros((k, 7, 8, 4), dtype=torch.float32, device=W.device)
    out[:, :, :7, :3] = W.detach().permute(0, 2, 3, 1)
    return out


class ResNetExtractorBase(torch.nn.Module):

    target_layer = 'res4'
    freeze_at = 'res2'
    _blocks = {50: [3, 4, 6, 3], 101: [3, 4, 23, 3]}

    def __init__(self, n_layers, remove_layers=None):
        super(ResNetExtractorBase, self).__init__()
        n = self._blocks[n_layers]
        self.conv1 = Convolution2D(3, 64, 7, 2, 3)
        self.bn1 = AffineChannel2D(64)
        self.res2 = BuildingBlock(n[0], 64, 64, 256, 1)
        self.res3 = BuildingBlock(n[1], 256, 128, 512, 2)
        self.res4 = BuildingBlock(n[2], 512, 256, 1024, 2)
        self.res5 = None
        if not remove_layers or 'res5' not in remove_layers:
            self.res5 = BuildingBlock(n[3], 1024, 512, 2048, 2)
        self._stem_cache = None
        self._prefetched = None       # (key, image batch, frozen-prefix activations, event)
        # optional {stage name: tensor hook}, fired when backward has passed the stage's output
        # (parallel.DataParallelGradSync polls its gradient buckets there)
        self.stage_hooks = {}

    def _stem(self, x):
        # conv1 and bn1 are frozen (examples/train_common.py:185-187): pack once per weight version
        W = self.conv1.W
        key = (W._version, W.data_ptr(), str(W.device))
        if self._stem_cache is None or self._stem_cache[0] != key:
            self._stem_cache = (key, pack_stem_filter(W))
        x4 = pad_image_nhwc4(x)
        return F.stem_conv(x4, self._stem_cache[1], self.conv1.b.detach(),
                           self.bn1.W.detach(), self.bn1.b.detach())

    @property
    def functions(self):
        return collections.OrderedDict([
            ('conv1', [self._stem]),
            ('pool1', [lambda x: F.max_pooling_2d(x, 3, stride=2, pad=1)]),
            ('res2', [self.res2]),
            ('res3', [self.res3]),
            ('res4', [self.res4]),
            ('res5', [self.res5]),
        ])

    # -- frozen prefix one step ahead ----------------------------------------------------------
    # conv1 .. freeze_at (res2) carry no gradient and their weights never change during training
    # (models/resnet_extractor.py:86-87 + examples/train_common.py:185-190), so their forward for
    # the NEXT image batch depends on nothing the current step produces.  `prefetch_frozen(x)`
    # queues it on a side stream; MaskRCNNTrainChain calls it from the hook that fires when the
    # head's and the RPN's backward are done — the batch-2 backbone backward that follows leaves
    # CUs idle (small-M GEMMs), and these equally small launches fill them.  `forward(x)` on the
    # SAME tensor then starts from the prefetched activations.  Values are those of an ordinary
    # forward (same kernels, same inputs); anything that writes the frozen weights must call
    # `drop_prefetched()` (the serializers do).
    def _frozen_prefix(self, x):
        h = x
        with torch.no_grad():
            for key, funcs in self.functions.items():
                for func in funcs:
                    h = func(h)
                if key == self.freeze_at:
                    break
        return h.detach()

    @staticmethod
    def _prefetch_key(x):
        return (id(x), x._version, x.data_ptr(), tuple(x.shape))

    def prefetch_frozen(self, x):
        if self.freeze_at is None or not isinstance(x, torch.Tensor) or not x.is_cuda:
            return
        if self._prefetched is not None and self._prefetched[0] == self._prefetch_key(x):
            return
        dev = x.device
        # the stream the deferred weight gradients use (idle between two proposal windows), NOT a
        # stream of its own: a fifth stream changes which HIP streams share a hardware queue
        # (GPU_MAX_HW_QUEUES = 4) and the proposal chain then queues behind the deferred weight
        # gradients — measured +2.4 ms under data parallelism, +6 ms with 5..8 queues (profiles/HISTORY.md 7a)
        from ..functions.conv import defer_stream, no_filter_cache
        side, main = defer_stream(dev), torch.cuda.current_stream(dev)
        side.wait_stream(main)
        # (the transformed-filter cache is ordered on the main stream only: bypassed here)
        with torch.cuda.stream(side), no_filter_cache():
            h = self._frozen_prefix(x)
            ready = torch.cuda.Event()
            ready.record(side)
        x.record_stream(side)
        self._prefetched = (self._prefetch_key(x), x, h, ready)

    def drop_prefetched(self):
        self._prefetched = None

    def forward(self, x):
        assert self.freeze_at is None or self.freeze_at in self.functions
        h = x
        frozen = self.freeze_at is not None
        pre, self._prefetched = self._prefetched, None
        skip_until = None
        if pre is not None and isinstance(x, torch.Tensor) and pre[0] == self._prefetch_key(x) \
                and torch.is_grad_enabled():
            main = torch.cuda.current_stream(x.device)
            main.wait_event(pre[3])
            h = pre[2]
            h.record_stream(main)
            skip_until, frozen = self.freeze_at, False
        for key, funcs in self.functions.items():
            if skip_until is not None:
                # a stage the prefetch already ran: nothing to compute; its hook has nothing to
                # fire on (no gradient below freeze_at), but target_layer still ends the walk —
                # possible only when it IS the last prefetched stage, whose output h holds
                if key == skip_until:
                    skip_until = None
                    if key == self.target_layer:
                        break
                else:
                    assert key != self.target_layer, \
                        'target_layer %r lies inside the prefetched frozen prefix' % (key,)
                continue
            for func in funcs:
                if frozen:
                    with torch.no_grad():
                        h = func(h)
                else:
                    h = func(h)
            if key == self.freeze_at:
                h = h.detach()          # Variable.unchain_backward()
                frozen = False
            hook = self.stage_hooks.get(key)
            if hook is not None and h.requires_grad:
                h.register_hook(hook)   # fires when backward has passed this stage
            if key == self.target_layer:
                break
        return h


class ResNet50Extractor(ResNetExtractorBase):
    def __init__(self, pretrained_model=None, remove_layers=None):
        super(ResNet50Extractor, self).__init__(50, remove_layers)


class ResNet101Extractor(ResNetExtractorBase):
    def __init__(self, pretrained_model=None, remove_layers=None):
        super(ResNet101Extractor, self).__init__(101, remove_layers)
