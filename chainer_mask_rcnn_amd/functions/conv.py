"""Convolution / deconvolution / linear ops on the fp32-MFMA implicit-GEMM kernel.

Host-side mirror of chainer's ``L.Convolution2D`` (+ the ``AffineChannel2D``,
residual add and ReLU that follow it inside chainer's Bottleneck blocks),
``L.Deconvolution2D(k=2, s=2)`` and ``L.Linear`` as the reference uses them
(/root/reference/chainer_mask_rcnn/models/region_proposal_network.py:75-80,
models/mask_rcnn_resnet.py:131-143).  Tensors carry the
reference's logical shapes — x (N,C,H,W), conv W (out,in,kh,kw), deconv W
(in,out,kh,kw), linear W (out,in) — stored channels-last, which is exactly the
NHWC / KRSC layout the HIP kernels consume.
"""
import torch

from .. import _lib
from .._lib import ConvDesc, EPI_BIAS, EPI_AFFINE, EPI_RESIDUAL, EPI_RELU, EPI_ACCUM, EPI_EXACT_SIGNS
from ._layout import nhwc, empty_nhwc


def _direct_grad(p):
    """Claim the right to WRITE ``p``'s gradient in place: true for an arena-backed parameter
    (optimizers.ParamArena) whose arena view is intact and that has not received a gradient yet
    in this step.  Otherwise the caller returns a fresh tensor and autograd accumulates it."""
    arena = getattr(p, '_arena', None)
    if arena is None:
        return False
    if arena.claim(p):
        return True
    join_wgrad_stream(p.device)      # an earlier in-place write may still be queued there
    return False


def conv_out_size(size, k, s, p):
    return (size + 2 * p - k) // s + 1


def make_desc(x_shape, w_shape, stride, pad):
    N, C, H, W = x_shape
    if len(w_shape) == 2:            # L.Linear weight (out, in) == 1x1 filter
        w_shape = (w_shape[0], w_shape[1], 1, 1)
    K, Cw, R, S = w_shape
    if Cw != C:
        raise ValueError('conv: input has %d channels, filter expects %d' % (C, Cw))
    return ConvDesc(N, H, W, C, K, R, S, stride, pad,
                    conv_out_size(H, R, stride, pad), conv_out_size(W, S, stride, pad))


def _colsum(g2d_ptr, M, C, out, device):
    ws = _lib.workspace(_lib.load().mrcnn_colsum_workspace_bytes(C), device, 'colsum')
    _lib.call('mrcnn_colsum', g2d_ptr, _lib.ptr(out), M, C, _lib.ptr(ws), _lib.stream_ptr())


def split_ws(device):
    """Scratch for the split-K leftover launches of small-M forward / dgrad GEMMs (cached per
    device and stream-ordered: safe because every user runs on the compute stream)."""
    return _lib.workspace(_lib.load().mrcnn_conv2d_split_workspace_bytes(), device, 'conv-split')


def epilogue_bwd(gy, y=None, scale=None):
    """g = gy * (y > 0) * scale[c] on NHWC tensors (any of y/scale may be None)."""
    N, C = gy.shape[0], gy.shape[1]
    M = gy.numel() // C
    g = torch.empty_like(gy)
    _lib.call('mrcnn_epilogue_bwd', _lib.ptr(gy), _lib.ptr(y), _lib.ptr(scale), _lib.ptr(g),
              M, C, _lib.stream_ptr())
    return g


_CONV_RECORDS_GRAPH = True

# Test hook: when set to a list, every ReLU output this module produces (fused conv epilogues, the
# three activations of each bottleneck of a fused stage, the deconvolution) is appended as
# (kind, tensor) in call order — tests/test_gpu_model.py reads the ReLU DECISIONS of the HIP path
# from it.  None (default): nothing is recorded or kept alive.
RELU_TAP = None


# ---- row-sparse backward of a 3x3 convolution (include/mrcnn_hip.h "row-sparse backward") -------
# The RPN's losses ignore every anchor but the <= 256 sampled ones per image, so the gradient
# reaching conv1's output is exactly zero outside their map positions; the train chain, which
# builds the anchor targets on the host, hands the positions to the RPN in a ``SparseRows``.
SPARSE_CONV_BACKWARD = True


class SparseRows(object):
    """Positions of a (N, H, W) map outside which the gradient arriving at a convolution's output
    is EXACTLY zero: ``rows`` int32 device tensor of sorted indices into N*H*W, ``lookup`` int32
    device tensor (N*H*W) position -> index into rows or -1, ``n`` = len(rows) (host int).
    One-shot and owned by ONE forward: the producer makes a new object per forward (the conv node
    records it), the backward that uses it clears it.  ``valid_for`` is the backward's guard: a hint
    whose tables do not describe the node's own (N, H, W) map is ignored (dense backward)."""

    def __init__(self):
        self.clear()

    def clear(self):
        self.rows = self.lookup = None
        self.n = 0

    def set(self, rows, lookup, n):
        self.rows, self.lookup, self.n = rows, lookup, int(n)

    def valid_for(self, d, device):
        """True when the tables can drive the row-sparse backward of a convolution with output
        map (d.N, d.P, d.Q) on ``device`` (shape / dtype / device checks only: no read-back)."""
        r, l = self.rows, self.lookup
        if r is None or l is None or self.n <= 0:
            return False
        size = d.N * d.P * d.Q
        return (r.dtype == torch.int32 and l.dtype == torch.int32 and r.device == device and
                l.device == device and r.is_contiguous() and l.is_contiguous() and
                r.numel() == self.n and self.n <= size and l.numel() == size)

    @staticmethod
    def host_tables(positions, size):
        """(rows int32 sorted unique, lookup int32 (size,)) NumPy arrays from map positions."""
        import numpy as np
        rows = np.unique(np.asarray(positions, np.int64)).astype(np.int32)
        lookup = np.full((int(size),), -1, np.int32)
        lookup[rows] = np.arange(len(rows), dtype=np.int32)
        return rows, lookup


_SPARSE_HINT = None


class sparse_output_grad(object):
    """``with sparse_output_grad(hint): y = conv2d(...)`` — the 3x3 / stride 1 / pad 1 convolutions
    recorded inside may take their backward from ``hint`` (a ``SparseRows`` filled before backward)."""

    def __init__(self, hint):
        self.hint = hint

    def __enter__(self):
        global _SPARSE_HINT
        self.prev, _SPARSE_HINT = _SPARSE_HINT, self.hint
        return self.hint

    def __exit__(self, *exc):
        global _SPARSE_HINT
        _SPARSE_HINT = self.prev
        return False


def _sparse3x3_backward(ctx, d, x, Wc, g, hint, need_x, need_w, need_b):
    """Backward of a 3x3 / stride 1 / pad 1 convolution whose output gradient ``g`` (already through
    ReLU) is zero outside ``hint.rows``: the 1x1 problem (N = rows, C = 9 C_in) on gathered patches."""
    dev, n = g.device, hint.n
    patches = torch.empty((n, 9 * d.C), dtype=torch.float32, device=dev)
    g_rows = torch.empty((n, d.K), dtype=torch.float32, device=dev)
    _lib.call('mrcnn_sparse3x3_gather', _lib.ptr(x), _lib.ptr(g), _lib.ptr(hint.rows), n,
              d.N, d.H, d.W, d.C, d.K, _lib.ptr(patches), _lib.ptr(g_rows), _lib.stream_ptr())
    d1 = ConvDesc(n, 1, 1, 9 * d.C, d.K, 1, 1, 1, 0, 1, 1)
    gx = gW = gb = None
    if need_w:
        gW = _wgrad_raw(d1, patches, g_rows, ctx.W_param, None, None)
    if need_x:
        gp = _dgrad_raw(d1, g_rows, Wc, None, None)          # (rows, 3, 3, C) patch gradients
        gx = empty_nhwc((d.N, d.C, d.H, d.W), dev)
        _lib.call('mrcnn_sparse3x3_scatter', _lib.ptr(gp), _lib.ptr(hint.lookup), d.N, d.H, d.W,
                  d.C, _lib.ptr(gx), _lib.stream_ptr())
    if need_b:
        b = ctx.b_param
        direct = _direct_grad(b)
        gbt = b.grad if direct else torch.empty((d.K,), dtype=torch.float32, device=dev)
        _colsum(_lib.ptr(g_rows), n, d.K, gbt, dev)
        gb = None if direct else gbt
    return gx, gW, gb


class _Conv2dFn(torch.autograd.Function):
    """y = relu?( affine?( conv(x, W) + b? ) + residual? )"""

    @staticmethod
    def forward(ctx, x, W, b, scale, shift, residual, stride, pad, relu):
        _lib.require_device(x, W)
        x = nhwc(x)
        Wc = nhwc(W) if W.dim() == 4 else W.contiguous()
        d = make_desc(x.shape, W.shape, stride, pad)
        flags = 0
        if b is not None:
            flags |= EPI_BIAS
        if scale is not None:
            flags |= EPI_AFFINE
        if residual is not None:
            residual = nhwc(residual)
            flags |= EPI_RESIDUAL
        if relu:
            flags |= EPI_RELU
        ctx.wino = uses_winograd(d) and residual is None and (b is None or scale is None)
        if ctx.wino and (WINOGRAD_TRAIN_FORWARD in (True, 'conv2d')
                         or not (_CONV_RECORDS_GRAPH and any(ctx.needs_input_grad))):
            y, _ = wino_fwd(x, Wc, d, scale, shift if scale is not None else b, relu,
                            cache_for=None if _CONV_RECORDS_GRAPH else W,
                            exact_signs=relu and _CONV_RECORDS_GRAPH and WINOGRAD_EXACT_SIGNS)
        else:
            y = empty_nhwc((d.N, d.K, d.P, d.Q), x.device)
            _lib.call('mrcnn_conv2d_fwd', ctx_desc(d), _lib.ptr(x), _lib.ptr(Wc), _lib.ptr(b),
                      _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(residual), _lib.ptr(y), flags,
                      _lib.ptr(split_ws(x.device)), _lib.stream_ptr())
        ctx.d = d
        ctx.relu = relu
        ctx.has_bias = b is not None
        ctx.has_res = residual is not None
        ctx.sparse_hint = _SPARSE_HINT if (d.R == 3 and d.S == 3 and stride == 1 and pad == 1 and
                                           scale is None and residual is None) else None
        ctx.W_param = W
        ctx.b_param = b
        ctx.save_for_backward(x, Wc, scale, y if relu else None)
        if RELU_TAP is not None and relu:
            RELU_TAP.append(('conv', y))
        return y

    @staticmethod
    def backward(ctx, gy):
        x, Wc, scale, y = ctx.saved_tensors
        d = ctx.d
        gy = nhwc(gy)
        need_x, need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], \
            ctx.has_bias and ctx.needs_input_grad[2]
        # through ReLU, then through the affine scale
        gr = epilogue_bwd(gy, y, None) if ctx.relu else gy
        g = epilogue_bwd(gr, None, scale) if scale is not None else gr
        M = d.N * d.P * d.Q
        gx = gW = gb = None
        hint = ctx.sparse_hint
        if hint is not None and SPARSE_CONV_BACKWARD and hint.valid_for(d, gy.device):
            gx, gW, gb = _sparse3x3_backward(ctx, d, x, Wc, g, hint, need_x, need_w, need_b)
            hint.clear()
            return gx, gW, gb, None, None, None, None, None, None
        if need_w:
            W = ctx.W_param
            if ctx.wino and WINOGRAD_WGRAD:
                gW = _wino_wgrad(d, x, None, g, W)
            elif USE_WGRAD_STREAM:
                gW = _wgrad_raw(d, x, g, W, None, None, wgrad_stream(gy.device))
            else:
                direct = _direct_grad(W)
                if direct:
                    gWt = W.grad
                elif W.dim() == 4:
                    gWt = empty_nhwc(tuple(W.shape), gy.device)
                else:
                    gWt = torch.empty_like(W)
                ws = _lib.workspace(_lib.load().mrcnn_conv2d_wgrad_workspace_bytes(ctx_desc(d)),
                                    gy.device, 'wgrad')
                _lib.call('mrcnn_conv2d_wgrad', ctx_desc(d), _lib.ptr(x), _lib.ptr(g),
                          _lib.ptr(gWt), _lib.ptr(ws), _lib.stream_ptr())
                gW = None if direct else gWt
        if need_x:
            gx = wino_dgrad(d, g, Wc) if ctx.wino and WINOGRAD_DGRAD else \
                _dgrad_raw(d, g, Wc, None, None)
        if need_b:
            b = ctx.b_param
            direct = _direct_grad(b)
            gbt = b.grad if direct else torch.empty((d.K,), dtype=torch.float32, device=gy.device)
            _colsum(_lib.ptr(g), M, d.K, gbt, gy.device)
            gb = None if direct else gbt
        gres = gr if ctx.has_res and ctx.needs_input_grad[5] else None
        return gx, gW, gb, None, None, gres, None, None, None


def ctx_desc(d):
    import ctypes
    return ctypes.byref(d)


def conv2d(x, W, b=None, stride=1, pad=0, scale=None, shift=None, residual=None, relu=False):
    """Fused convolution: ``relu(affine(conv(x,W)+b) + residual)`` with each stage optional.

    ``scale``/``shift`` are the AffineChannel2D ``W``/``b`` (frozen: no gradient is
    produced for them — the reference computes but never uses it,
    examples/train_common.py:188-190).
    """
    global _CONV_RECORDS_GRAPH
    _CONV_RECORDS_GRAPH = torch.is_grad_enabled()     # see _STAGE_RECORDS_GRAPH
    try:
        return _Conv2dFn.apply(x, W, b, scale, shift, residual, stride, pad, relu)
    finally:
        _CONV_RECORDS_GRAPH = True


class _StemFn(torch.autograd.Function):
    """conv1 7x7/2 pad 3 + bias + affine + ReLU, forward only (frozen stem)."""

    @staticmethod
    def forward(ctx, x4, w784, b, scale, shift):
        _lib.require_device(x4, w784)
        N, H, W_, four = x4.shape
        assert four == 4 and x4.is_contiguous()
        K = w784.shape[0]
        P, Q = conv_out_size(H, 7, 2, 3), conv_out_size(W_, 7, 2, 3)
        y = empty_nhwc((N, K, P, Q), x4.device)
        flags = EPI_RELU | (EPI_BIAS if b is not None else 0) | \
            (EPI_AFFINE if scale is not None else 0)
        _lib.call('mrcnn_conv_stem_fwd', _lib.ptr(x4), _lib.ptr(w784), _lib.ptr(b),
                  _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(y), N, H, W_, K, flags,
                  _lib.stream_ptr())
        return y

    @staticmethod
    def backward(ctx, gy):
        raise _lib.MrcnnHipError(
            'the ResNet stem is frozen (unchain_backward at res2, '
            'models/resnet_extractor.py:86-87): no backward is implemented')


def stem_conv(x4, w784, b, scale, shift):
    return _StemFn.apply(x4, w784, b, scale, shift)


# The deconvolution's forward as a forward-form GEMM on the transposed filter (split-operand kernels)
# instead of the K-strided data-gradient form (fp32 MFMA): see mrcnn_deconv2x2s2_fwd_wt.
DECONV_FORWARD_FORM = True


class _Deconv2x2Fn(torch.autograd.Function):
    """L.Deconvolution2D(in, out, 2, stride=2) (+ bias, + ReLU)."""

    @staticmethod
    def forward(ctx, x, W, b, relu):
        _lib.require_device(x, W)
        x = nhwc(x)
        Wc = nhwc(W)     # logical (C, K, 2, 2) -> physical (C, 2, 2, K)
        N, C, H, Wd = x.shape
        K = W.shape[1]
        if tuple(W.shape) != (C, K, 2, 2):
            raise ValueError('deconv: filter must be (in, out, 2, 2), got %s' % (tuple(W.shape),))
        y = empty_nhwc((N, K, 2 * H, 2 * Wd), x.device)
        flags = (EPI_BIAS if b is not None else 0) | (EPI_RELU if relu else 0)
        if DECONV_FORWARD_FORM and GEMM_ARITHMETIC == 'split_bf16x3':
            # forward form on the transposed filter (4K, C): both operands K-contiguous -> the
            # split-operand kernels; the K-strided form below runs on fp32 MFMA
            wT = _lib.workspace(4 * C * 4 * K, x.device, 'deconv-wT').view(torch.float32)
            _lib.call('mrcnn_filter_flip_transpose', _lib.ptr(Wc), _lib.ptr(wT), C, 1, 1, 4 * K, None,
                      _lib.stream_ptr())
            _lib.call('mrcnn_deconv2x2s2_fwd_wt', _lib.ptr(x), _lib.ptr(wT), _lib.ptr(b), _lib.ptr(y),
                      N, H, Wd, C, K, flags, _lib.stream_ptr())
        else:
            _lib.call('mrcnn_deconv2x2s2_fwd', _lib.ptr(x), _lib.ptr(Wc), _lib.ptr(b), _lib.ptr(y),
                      N, H, Wd, C, K, flags, _lib.stream_ptr())
        ctx.dims = (N, H, Wd, C, K)
        ctx.relu = relu
        ctx.W_param, ctx.b_param = W, b
        ctx.save_for_backward(x, Wc, y if relu else None)
        if RELU_TAP is not None and relu:
            RELU_TAP.append(('deconv', y))
        return y

    @staticmethod
    def backward(ctx, gy):
        x, Wc, y = ctx.saved_tensors
        N, H, Wd, C, K = ctx.dims
        gy = nhwc(gy)
        g = epilogue_bwd(gy, y, None) if ctx.relu else gy
        gx = gW = gb = None
        if ctx.needs_input_grad[0]:
            gx = empty_nhwc((N, C, H, Wd), gy.device)
            _lib.call('mrcnn_deconv2x2s2_dgrad', _lib.ptr(g), _lib.ptr(Wc), _lib.ptr(gx),
                      N, H, Wd, C, K, _lib.stream_ptr())
        if ctx.needs_input_grad[1]:
            W = ctx.W_param
            direct = _direct_grad(W)
            gWt = W.grad if direct else empty_nhwc(tuple(W.shape), gy.device)
            ws = _lib.workspace(
                _lib.load().mrcnn_deconv2x2s2_wgrad_workspace_bytes(N, H, Wd, C, K),
                gy.device, 'wgrad')
            _lib.call('mrcnn_deconv2x2s2_wgrad', _lib.ptr(x), _lib.ptr(g), _lib.ptr(gWt),
                      N, H, Wd, C, K, _lib.ptr(ws), _lib.stream_ptr())
            gW = None if direct else gWt
        if ctx.b_param is not None and ctx.needs_input_grad[2]:
            b = ctx.b_param
            direct = _direct_grad(b)
            gbt = b.grad if direct else torch.empty((K,), dtype=torch.float32, device=gy.device)
            _colsum(_lib.ptr(g), N * 4 * H * Wd, K, gbt, gy.device)
            gb = None if direct else gbt
        return gx, gW, gb, None


def deconv2x2s2(x, W, b=None, relu=False):
    return _Deconv2x2Fn.apply(x, W, b, relu)


def linear(x, W, b=None):
    """L.Linear: y = x.reshape(N,-1) @ W.T + b, as a 1x1 convolution on (N,C,1,1)."""
    n = x.shape[0]
    x4 = x.reshape(n, -1, 1, 1)
    y = conv2d(x4, W, b)
    return y.reshape(n, W.shape[0])


# ---------------------------------------------------------------------------------------
# Whole-bottleneck op: chainer BottleneckA / BottleneckB as ONE autograd
# node.  Forward = 3 (4) fused implicit-GEMM launches; backward = 3 (4) dgrad + 3 (4) wgrad
# launches and NOTHING else: the ReLU masks and affine scales are applied while the
# incoming gradient is staged into LDS, and the identity-shortcut gradient is added in the
# dgrad epilogue, so there is no elementwise pass and no gradient-accumulation kernel.
# ---------------------------------------------------------------------------------------

# Measured on MI355X (round 1): queueing wgrads on a second stream did not overlap with the
# dgrads in practice (68.9 vs 68.1 ms per step), so the single-stream order is the default.
USE_WGRAD_STREAM = False


def _fwd_raw(x, Wc, d, scale, shift, residual, relu):
    flags = (EPI_AFFINE if scale is not None else 0) | (EPI_RESIDUAL if residual is not None else 0) \
        | (EPI_RELU if relu else 0)
    y = empty_nhwc((d.N, d.K, d.P, d.Q), x.device)
    _lib.call('mrcnn_conv2d_fwd', ctx_desc(d), _lib.ptr(x), _lib.ptr(Wc), None, _lib.ptr(scale),
              _lib.ptr(shift), _lib.ptr(residual), _lib.ptr(y), flags, _lib.ptr(split_ws(x.device)),
              _lib.stream_ptr())
    return y


USE_TRANSPOSED_DGRAD = True


def _flip_transpose(Wc, d, row_scale, out):
    _lib.call('mrcnn_filter_flip_transpose', _lib.ptr(Wc), _lib.ptr(out), d.K, d.R, d.S, d.C,
              _lib.ptr(row_scale), _lib.stream_ptr())
    return out


def _stage_transposes(blocks, scales, device):
    """Flipped / transposed filters of every stride-1 convolution of a stage, built by ONE
    launch into one buffer.  ``blocks``: ((d1,d2,d3,d4), (W1,W2,W3,W4), _) per bottleneck,
    ``scales``: (s3, s4) per bottleneck (folded into conv3 / conv4).  Returns a dict
    {'1','2','3','4'} -> flat tensor per block."""
    import ctypes
    jobs = []
    for bi, ((d1, d2, d3, d4), (W1, W2, W3, W4), _) in enumerate(blocks):
        s3, s4 = scales[bi]
        for key, W, d, sc in (('1', W1, d1, None), ('2', W2, d2, None), ('3', W3, d3, s3),
                              ('4', W4, d4, s4)):
            if W is not None and _uses_transposed_dgrad(d) and not uses_winograd(d):
                jobs.append((bi, key, nhwc(W), d, sc))
    out = [dict() for _ in blocks]
    if not jobs:
        return out
    total = sum(j[2].numel() for j in jobs)
    buf = torch.empty((total,), dtype=torch.float32, device=device)
    n = len(jobs)
    vp, ci = ctypes.c_void_p * n, ctypes.c_int * n
    w, wT, sc = vp(), vp(), vp()
    K, R, S, C = ci(), ci(), ci(), ci()
    off = 0
    keep = []
    for i, (bi, key, Wc, d, scale) in enumerate(jobs):
        t = buf[off:off + Wc.numel()]
        off += Wc.numel()
        out[bi][key] = t
        keep.append(Wc)
        w[i], wT[i] = Wc.data_ptr(), t.data_ptr()
        sc[i] = scale.data_ptr() if scale is not None else None
        K[i], R[i], S[i], C[i] = d.K, d.R, d.S, d.C
    _lib.call('mrcnn_filter_flip_transpose_batched', n, w, wT, K, R, S, C, sc, _lib.stream_ptr())
    return out


# Strided 1x1 / pad 0 data gradients (res3.a / res4.a conv1 and conv4) in forward form on the
# transposed filter as well: dense gather of gy, rows scattered to the strided pixels of a zero-filled
# gx (split-operand kernels; the K-strided form runs on fp32 MFMA).
STRIDED_DGRAD_FORWARD_FORM = True


def _uses_transposed_dgrad(d):
    if not USE_TRANSPOSED_DGRAD:
        return False
    if d.stride == 1:
        return d.R == d.S
    return STRIDED_DGRAD_FORWARD_FORM and d.R == 1 and d.S == 1 and d.pad == 0


def _dgrad_raw(d, g, Wc, mask_y, in_scale, res_g=None, res_y=None, out=None, accum=False,
               out_mask_y=None, out_scale=None, fold_scale=None, wT=None):
    """gx = dgrad(g') with the fused pieces of include/mrcnn_hip.h "Extended backward entry
    points": consumer-side ``mask_y`` / ``in_scale``, producer-side ``out_mask_y`` /
    ``out_scale``, shortcut gradient ``res_g`` (masked by ``res_y`` if given).  ``fold_scale``
    is a per-output-channel scale of the incoming gradient folded into the filter (stride 1)
    or, for the strided kernel, passed as ``in_scale``."""
    gx = out if out is not None else empty_nhwc((d.N, d.C, d.H, d.W), g.device)
    if _uses_transposed_dgrad(d):
        # forward-form dgrad on the flipped, transposed filter; ``wT`` = already built (with
        # fold_scale applied), else rebuilt here (it moves 2 x the filter bytes)
        if wT is None:
            wT = _flip_transpose(Wc, d, fold_scale,
                                 _lib.workspace(4 * d.K * d.R * d.S * d.C, g.device, 'wT'))
        _lib.call('mrcnn_conv2d_dgrad_wt', ctx_desc(d), _lib.ptr(g), _lib.ptr(wT), _lib.ptr(gx),
                  EPI_ACCUM if accum else 0, _lib.ptr(mask_y), _lib.ptr(in_scale),
                  _lib.ptr(res_g), _lib.ptr(res_y), _lib.ptr(out_mask_y), _lib.ptr(out_scale),
                  _lib.ptr(split_ws(g.device)), _lib.stream_ptr())
        return gx
    if fold_scale is not None:
        assert in_scale is None
        in_scale = fold_scale
    _lib.call('mrcnn_conv2d_dgrad_ex', ctx_desc(d), _lib.ptr(g), _lib.ptr(Wc), _lib.ptr(gx),
              EPI_ACCUM if accum else 0, _lib.ptr(mask_y), _lib.ptr(in_scale), _lib.ptr(res_g),
              _lib.ptr(res_y), _lib.ptr(out_mask_y), _lib.ptr(out_scale),
              _lib.ptr(split_ws(g.device)), _lib.stream_ptr())
    return gx


# ---- Winograd F(4x4,3x3) path (csrc/conv_winograd.h) ------------------------------------------
# The RoI head's 3x3 convolutions run on ~1000 maps of 7x7: padded to 8x8 that is four 4x4
# output tiles per map, 144 multiply-adds per (map, c, k) instead of the direct form's 441.
# Selected for 3x3 / stride 1 / pad 1 layers over many small maps; the backbone's large maps keep
# the implicit-GEMM kernel.  fp32 throughout; error 3.4e-6 of the tensor scale (direct: 3.5e-7),
# parity tolerance 1e-4.
USE_WINOGRAD = True
# Which layers: 3x3 / stride 1 / pad 1 with enough work per frequency plane for the 36 batched
# GEMMs to fill the GPU — the RoI head's res5 (1024 maps of 7x7, 512 -> 512), the RPN's conv1
# (1024 -> 1024) and, at inference batch sizes, res4 (256 -> 256).  Narrow layers (C < 256) and
# the two-image res4 maps of a train step stay on the implicit-GEMM kernel: their GEMMs would
# be 2-8 K slices deep and the transform passes would cost what the MFMAs save.
import os as _os
WINOGRAD_MIN_CHANNELS = int(_os.environ.get('MRCNN_WINO_MIN_CH', 256))
WINOGRAD_MIN_WORK = int(_os.environ.get('MRCNN_WINO_MIN_WORK', 1 << 27))          # tiles x C x K
# Which passes take the Winograd route.  Backward-data and backward-filter always do: their
# extra rounding (3e-6 of the gradient tensor's scale) is invisible next to the fp32 floor of
# the whole-graph gradients (tools/grad_floor.py: the per-layer errors against the float64 graph
# are the same with and without).  The FORWARD of a recorded (training) graph is decided per
# layer kind, by the same measurement on the random-init R-101 of tests/test_gpu_model.py:
#   * F.conv2d layers (the RPN's conv1) take it: worst layer 3.6e-4 of entries beyond 1e-4 of
#     the tensor scale, max 5.6e-4 — the same figures as with the direct forward;
#   * the fused stages (res5 in the RoI head) do NOT.  The forward difference itself is benign
#     (Winograd vs direct output: rms 9e-7, max 3e-6 of the output scale at every head layer of
#     both test nets, activations max/rms 7-11: tools/exp/head_activation_stats.py), but it is
#     ten times the direct kernel's rounding, and every ReLU DOWNSTREAM of the layer (conv3 +
#     shortcut, the next blocks) sees it: ten times as many decisions of units sitting at zero
#     flip.  On the R-101 instance one such flip behind res5.b1.conv2 carries enough gradient
#     that 69 of the 512 rows of head.res5.b1.conv{1,2}.W (and a few rows of most backbone
#     layers) moved beyond 1e-4 of the tensor scale — 0.2 % of the entries where the parity
#     test allows 0.1 %.  (R-50, and the same R-101 with res5's residual branches damped like
#     res4's, pass with the forward routed: it is a lottery, lost once.)
# A routed forward inside a recorded graph asks for MRCNN_EPI_EXACT_SIGNS: outputs within the
# propagated rounding bound of zero (5 in 10^5) are recomputed as direct dot products, so the
# layer's OWN ReLU decisions agree with float64 at least as often as the direct kernel's
# (0-1 disagreements per 25 M outputs, direct kernel 4-6, plain Winograd 12-22:
# tools/exp/wino_signs.py).  That settles the RPN's conv1, behind whose ReLU there is only a 1x1
# convolution and the losses; it cannot help the ReLUs further down a fused stage.
# Without a graph (inference) every routed layer's forward takes it: only the per-op tolerance
# applies there and it holds with a 30x margin.
WINOGRAD_TRAIN_FORWARD = True         # False / 'conv2d' / 'stage' / True (both; default since round 3)
WINOGRAD_EXACT_SIGNS = True          # recorded graphs: ReLU decisions recomputed directly near zero
WINOGRAD_DGRAD = True        # developer switches (error attribution, A/B timing)
WINOGRAD_WGRAD = True


def uses_winograd(d):
    if not (USE_WINOGRAD and d.R == 3 and d.S == 3 and d.stride == 1 and d.pad == 1):
        return False
    tiles = d.N * ((d.H + 3) // 4) * ((d.W + 3) // 4)
    return min(d.C, d.K) >= WINOGRAD_MIN_CHANNELS and tiles * d.C * d.K >= WINOGRAD_MIN_WORK


def _wino_ws(d, device, tag='wino'):
    return _lib.workspace(_lib.load().mrcnn_conv3x3_wino_workspace_bytes(ctx_desc(d)), device, tag)


# Transformed filters of inference calls, keyed by the filter tensor: valid while the tensor has
# not been written (torch bumps ``_version`` on every in-place update, the optimizer's included).
_wino_u_cache = {}


# ---- arithmetic of the convolution GEMMs ----------------------------------------------------
# 'split_bf16x3' (default since round 4): every fp32 operand element is split EXACTLY into three
# bf16 values (8 + 8 + 8 significand bits) while it is staged, and a K step runs six bf16 MFMAs
# (hi*hi, hi*mid, mid*hi, mid*mid, hi*lo, lo*hi) with fp32 accumulation; the three dropped cross
# terms (mid*lo, lo*mid, lo*lo) are <= 2^-24 of a product each, <= 2^-23 in total — the size of ONE
# fp32 rounding — so the result carries the error of an fp32 GEMM (tests/test_gpu_split_bf16.py measures it against float64 next to the fp32-MFMA
# kernels': smaller on every case) at 2.7x the matrix-pipe rate (gfx950 has no TF32 / xf32 and its
# fp32 MFMA runs at 1/16 of the bf16 rate).  Covers the forward-form kernels (forward, stride-1 data
# gradient, the Winograd per-frequency GEMMs) and the 128x128 weight gradient; the strided data
# gradient, the position-major and the 64x64 weight gradient stay on fp32 MFMA.
# 'fp32': v_mfma_f32_32x32x2_f32 everywhere (the default up to round 3; bench.py reports it as
# `fp32_mfma`).
DEFAULT_GEMM_ARITHMETIC = 'split_bf16x3'
GEMM_ARITHMETIC = DEFAULT_GEMM_ARITHMETIC


def set_gemm_arithmetic(kind):
    """Select 'split_bf16x3' (default) or 'fp32' for the convolution GEMM kernels (process-wide).

    Operand range of 'split_bf16x3' (bf16 has fp32's exponent range, so nothing is rescaled):
    * finite elements up to bf16's largest value (3.3895e38; |x| <= 0.99 FLT_MAX is safe) split
      exactly, whatever mix of magnitudes a K row or a tile holds (per element, no shared exponent);
    * an element that is NaN, +-inf or finite but beyond that value (it rounds to +-inf in bf16)
      makes EVERY output that reads it NaN — fp32 multiplication would give +-inf for the
      infinities; outputs that do not read it are unaffected;
    * parts of an element below 2^-126 may be flushed to zero: the error of an element's
      representation is < 3 x 2^-126 in absolute terms (relative to elements below ~1e-33 that is
      more than an fp32 rounding).
    Asserted by tests/test_gpu_split_bf16.py (device) and tests/test_split_arithmetic_cpu.py (model)."""
    global GEMM_ARITHMETIC
    if kind not in ('fp32', 'split_bf16x3'):
        raise ValueError("gemm arithmetic must be 'fp32' or 'split_bf16x3', got %r" % (kind,))
    _lib.set_tuning('split_bf16', 3 if kind == 'split_bf16x3' else 0)
    GEMM_ARITHMETIC = kind


def weights_changed():
    """Public invalidation call: parameters were written outside torch's version tracking — the
    SGD kernel updates the flat arena through a raw pointer; user code that writes through
    ``p.data`` (``p.data.copy_``, ``np.copyto`` on a mapped array, ``load_state_dict(assign=
    True)``) does not bump ``_version`` either.  Drops every cached transformed filter; the
    serializers call it, and so must any other out-of-band writer."""
    _wino_u_cache.clear()
    _proj_cache.clear()


# The cache has no cross-stream ordering: an entry is produced and read on the caller's current
# stream and freed by ``weights_changed`` (every SGD step).  Work queued on a SIDE stream (the
# frozen-prefix prefetch of models/resnet_extractor.py) therefore bypasses it.
_WINO_CACHE_BYPASS = 0


class no_filter_cache(object):
    """``with no_filter_cache(): ...`` — Winograd convolutions inside transform their filter per call
    instead of reading / filling the per-parameter cache (for work on a stream other than the one
    the cache's entries live on)."""

    def __enter__(self):
        global _WINO_CACHE_BYPASS
        _WINO_CACHE_BYPASS += 1

    def __exit__(self, *exc):
        global _WINO_CACHE_BYPASS
        _WINO_CACHE_BYPASS -= 1
        return False


def _cached_filter_transform(W, Wc, d):
    key = id(W)
    hit = _wino_u_cache.get(key)
    # identity + version + storage address: ``p.data = t`` / ``assign=True`` loads swap the
    # storage without touching the version counter
    if hit is not None and hit[0] is W and hit[1] == W._version and hit[3] == W.data_ptr() \
            and hit[2].device == Wc.device:
        return hit[2]
    u = torch.empty((_lib.load().mrcnn_conv3x3_wino_u_bytes(ctx_desc(d)) // 4,),
                    dtype=torch.float32, device=Wc.device)
    _lib.call('mrcnn_conv3x3_wino_filter', ctx_desc(d), _lib.ptr(Wc), _lib.ptr(u), _lib.stream_ptr())
    _wino_u_cache[key] = (W, W._version, u, W.data_ptr())
    return u


def wino_fwd(x, Wc, d, scale, shift, relu, keep_v=False, cache_for=None, exact_signs=False):
    """y = relu?(affine?(conv3x3(x))) — ``scale`` None with a ``shift``: plain bias — and, with
    ``keep_v``, the transformed input (36, tiles, C) for the weight gradient.  ``cache_for``: the
    parameter tensor ``Wc`` was taken from; its transformed filter is then built once and reused
    until the parameter is written (inference).  ``exact_signs``: outputs within the Winograd
    rounding bound of zero are recomputed directly (MRCNN_EPI_EXACT_SIGNS)."""
    flags = (EPI_AFFINE if scale is not None else (EPI_BIAS if shift is not None else 0)) \
        | (EPI_RELU if relu else 0) | (EPI_EXACT_SIGNS if exact_signs else 0)
    y = empty_nhwc((d.N, d.K, d.P, d.Q), x.device)
    v = None
    if keep_v:
        v = torch.empty((_lib.load().mrcnn_conv3x3_wino_v_bytes(ctx_desc(d)) // 4,),
                        dtype=torch.float32, device=x.device)
    u = _cached_filter_transform(cache_for, Wc, d) \
        if cache_for is not None and not _WINO_CACHE_BYPASS else None
    _lib.call('mrcnn_conv3x3_wino_fwd', ctx_desc(d), _lib.ptr(x), _lib.ptr(Wc), _lib.ptr(u),
              _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(y), flags, _lib.ptr(v),
              _lib.ptr(_wino_ws(d, x.device)), _lib.stream_ptr())
    return y, v


def wino_dgrad(d, g, Wc, fold_scale=None, out_scale=None, out_mask_y=None):
    gx = empty_nhwc((d.N, d.C, d.H, d.W), g.device)
    _lib.call('mrcnn_conv3x3_wino_dgrad', ctx_desc(d), _lib.ptr(g), _lib.ptr(Wc),
              _lib.ptr(fold_scale), _lib.ptr(gx), _lib.ptr(out_scale), _lib.ptr(out_mask_y),
              _lib.ptr(_wino_ws(d, g.device)), _lib.stream_ptr())
    return gx


def wino_wgrad_into(d, x, v, g, gW, row_scale=None, tag='wino'):
    """gW = backward-filter from ``g`` and exactly one of ``x`` (raw input) / ``v`` (the
    transformed input a forward with ``keep_v`` returned)."""
    _lib.call('mrcnn_conv3x3_wino_wgrad', ctx_desc(d), _lib.ptr(x), _lib.ptr(v), _lib.ptr(g),
              _lib.ptr(gW), _lib.ptr(row_scale), _lib.ptr(_wino_ws(d, g.device, tag)),
              _lib.stream_ptr())


# ---- weight-gradient side stream -----------------------------------------------------------
# dgrad and wgrad of a convolution are independent given the incoming gradient.  The wgrad
# launches of a bottleneck go to a second HIP stream so that their workgroups fill the CUs
# that the tail round of the concurrently running dgrad leaves idle (and vice versa): with
# 128x128 tiles and 512 resident workgroups a single kernel wastes up to a round at its end.
_side_streams = {}


def wgrad_stream(device):
    key = str(device)
    if key not in _side_streams:
        _side_streams[key] = torch.cuda.Stream(device=device)
    return _side_streams[key]


def wgrad_stream_if_used(device):
    """The weight-gradient side stream of ``device`` if one has been created, else None."""
    return _side_streams.get(str(device))


def join_wgrad_stream(device=None):
    """Make the current stream wait for every weight gradient queued on the side stream
    (call before reading gradients: optimizer step, all-reduce)."""
    for key, st in _side_streams.items():
        if device is None or key == str(device):
            torch.cuda.current_stream(st.device).wait_stream(st)


# ---- weight gradients deferred into the next step's proposal window --------------------------
# Between the RPN convolution and the RoI head of a train step the GPU is nearly idle: the
# proposal kernels (top-k, NMS scan: two workgroups) run, then the host samples the RoIs
# (~2 ms together).  The RoI head's weights are not read before that window has passed, so the
# weight gradients of some head layers — and the SGD update of exactly those parameters — can be
# held back at the end of a step and run INSIDE the next step's window on a second stream, where
# they cost nothing (optimizers.MomentumSGD.defer_weight_gradients; same gradients, same update,
# applied before the parameter is read again: results are bit-identical).
class DeferQueue(object):
    def __init__(self, params):
        self.ids = set(id(p) for p in params)
        self.jobs = []


_DEFER = None          # set by MomentumSGD.update around backward
_defer_streams = {}


def defer_stream(device):
    key = str(device)
    if key not in _defer_streams:
        _defer_streams[key] = torch.cuda.Stream(device=device)
    return _defer_streams[key]


def run_deferred_wgrads(jobs):
    """Launch the held-back weight gradients on the current stream (the caller selects it)."""
    for d, x, g, gW, mask_y, in_scale, row_scale in jobs:
        if mask_y is _WINO:
            wino_wgrad_into(d, x, in_scale, g, gW, row_scale, tag='wino-defer')
            continue
        ws = _lib.workspace(_lib.load().mrcnn_conv2d_wgrad_workspace_bytes(ctx_desc(d)),
                            g.device, 'wgrad-defer')
        _lib.call('mrcnn_conv2d_wgrad_ex', ctx_desc(d), _lib.ptr(x), _lib.ptr(g), _lib.ptr(gW),
                  _lib.ptr(ws), _lib.ptr(mask_y), _lib.ptr(in_scale), _lib.ptr(row_scale),
                  _lib.stream_ptr())


_WINO = object()       # marker in the mask_y slot of a deferred job: Winograd weight gradient


def _wino_wgrad(d, x, v, g, W, row_scale=None):
    """Weight gradient on the Winograd route from the raw input ``x`` or the kept transformed
    input ``v`` (the other is None); same gradient-ownership rules as _wgrad_raw."""
    direct = _direct_grad(W)
    if direct and _DEFER is not None and id(W) in _DEFER.ids:
        _DEFER.jobs.append((d, x, g, W.grad, _WINO, v, row_scale))    # v rides in the in_scale slot
        return None
    gW = W.grad if direct else empty_nhwc(tuple(W.shape), g.device)
    wino_wgrad_into(d, x, v, g, gW, row_scale)
    return None if direct else gW


def _wgrad_raw(d, x, g, W, mask_y, in_scale, side=None, row_scale=None):
    """Returns the tensor autograd should see for W (None when written in place).  With
    ``side`` (a stream) the launch is queued there, ordered after everything queued so far
    on the current stream; only arena-backed (direct) gradients may use it."""
    direct = _direct_grad(W)
    if direct and _DEFER is not None and id(W) in _DEFER.ids:
        _DEFER.jobs.append((d, x, g, W.grad, mask_y, in_scale, row_scale))
        return None
    gW = W.grad if direct else empty_nhwc(tuple(W.shape), g.device)
    if side is not None and direct:
        side.wait_stream(torch.cuda.current_stream(g.device))
        for t in (x, g, mask_y):
            if t is not None:
                t.record_stream(side)
        with torch.cuda.stream(side):
            ws = _lib.workspace(_lib.load().mrcnn_conv2d_wgrad_workspace_bytes(ctx_desc(d)),
                                g.device, 'wgrad-side')
            _lib.call('mrcnn_conv2d_wgrad_ex', ctx_desc(d), _lib.ptr(x), _lib.ptr(g),
                      _lib.ptr(gW), _lib.ptr(ws), _lib.ptr(mask_y), _lib.ptr(in_scale),
                      _lib.ptr(row_scale), _lib.stream_ptr())
        return None
    ws = _lib.workspace(_lib.load().mrcnn_conv2d_wgrad_workspace_bytes(ctx_desc(d)),
                        g.device, 'wgrad')
    _lib.call('mrcnn_conv2d_wgrad_ex', ctx_desc(d), _lib.ptr(x), _lib.ptr(g), _lib.ptr(gW),
              _lib.ptr(ws), _lib.ptr(mask_y), _lib.ptr(in_scale), _lib.ptr(row_scale),
              _lib.stream_ptr())
    return None if direct else gW


class _BottleneckFn(torch.autograd.Function):

    @staticmethod
    def forward(ctx, x, W1, s1, b1, W2, s2, b2, W3, s3, b3, W4, s4, b4, stride):
        _lib.require_device(x, W1)
        x = nhwc(x)
        d1 = make_desc(x.shape, W1.shape, stride, 0)
        h1 = _fwd_raw(x, nhwc(W1), d1, s1, b1, None, True)
        d2 = make_desc(h1.shape, W2.shape, 1, 1)
        h2 = _fwd_raw(h1, nhwc(W2), d2, s2, b2, None, True)
        d4 = None
        if W4 is not None:
            d4 = make_desc(x.shape, W4.shape, stride, 0)
            shortcut = _fwd_raw(x, nhwc(W4), d4, s4, b4, None, False)
        else:
            shortcut = x
        d3 = make_desc(h2.shape, W3.shape, 1, 0)
        y = _fwd_raw(h2, nhwc(W3), d3, s3, b3, shortcut, True)
        ctx.descs = (d1, d2, d3, d4)
        ctx.params = (W1, W2, W3, W4)
        ctx.save_for_backward(x, h1, h2, y, s1, s2, s3, s4)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, h1, h2, y, s1, s2, s3, s4 = ctx.saved_tensors
        d1, d2, d3, d4 = ctx.descs
        W1, W2, W3, W4 = ctx.params
        gy = nhwc(gy)
        ng = ctx.needs_input_grad
        gW1 = gW2 = gW3 = gW4 = gx = None
        side = wgrad_stream(gy.device) if USE_WGRAD_STREAM else None
        # conv3 <- relu/affine(bn3) of the block output; each wgrad is queued on the side
        # stream right after the dgrad that produces its incoming gradient
        if ng[7]:
            gW3 = _wgrad_raw(d3, h2, gy, W3, y, s3, side)
        if W4 is not None and ng[10]:
            gW4 = _wgrad_raw(d4, x, gy, W4, y, s4, side)
        gh2 = _dgrad_raw(d3, gy, nhwc(W3), y, s3)
        if ng[4]:
            gW2 = _wgrad_raw(d2, h1, gh2, W2, h2, s2, side)
        gh1 = _dgrad_raw(d2, gh2, nhwc(W2), h2, s2)
        if ng[1]:
            gW1 = _wgrad_raw(d1, x, gh1, W1, h1, s1, side)
        if ng[0]:
            if W4 is None:
                # identity shortcut: gx = dgrad(conv1) + gy * (y > 0), added in the epilogue
                gx = _dgrad_raw(d1, gh1, nhwc(W1), h1, s1, res_g=gy, res_y=y)
            else:
                gx = _dgrad_raw(d1, gh1, nhwc(W1), h1, s1)
                _dgrad_raw(d4, gy, nhwc(W4), y, s4, out=gx, accum=True)
        return (gx, gW1, None, None, gW2, None, None, gW3, None, None, gW4, None, None, None)


def bottleneck(x, conv1, bn1, conv2, bn2, conv3, bn3, conv4=None, bn4=None, stride=1):
    """relu(bn3(conv3(relu(bn2(conv2(relu(bn1(conv1 x))))))) + shortcut) with
    shortcut = bn4(conv4 x) (BottleneckA) or x (BottleneckB); stride lives in conv1/conv4."""
    return _BottleneckFn.apply(
        x, conv1.W, bn1.W, bn1.b, conv2.W, bn2.W, bn2.b, conv3.W, bn3.W, bn3.b,
        None if conv4 is None else conv4.W, None if bn4 is None else bn4.W,
        None if bn4 is None else bn4.b, stride)


# ---------------------------------------------------------------------------------------
# Whole-stage op: chainer BuildingBlock (BottleneckA + n x BottleneckB) as ONE autograd node.
# Same launches as a chain of _BottleneckFn, but every gradient tensor inside the stage is
# written ALREADY multiplied by the ReLU mask (and affine scale) of the conv that consumes it
# — in the epilogue of the dgrad that produces it — so none of the stage's backward GEMMs
# stages a mask: they all run the plain 168-register kernels, three workgroups per CU.  Only
# the gradient entering the stage from autograd takes one elementwise masking pass.
# ---------------------------------------------------------------------------------------
# Weight gradients of the backbone's small-M layers (<= 40 000 output pixels: res3 / res4) are
# queued on a second HIP stream: such a launch has only 2-4 workgroups per CU, so the wgrad and
# the next dgrad share the GPU (same-box A/B: 54.4 -> 54.0 ms per step).  For the large res5
# launches the same trick measured no overlap (USE_WGRAD_STREAM above).
SMALL_WGRAD_SIDE_STREAM = True
SMALL_WGRAD_MAX_PIXELS = int(_os.environ.get('MRCNN_SIDE_WGRAD_MAX_PIXELS', 40000))
# Build the transposed filters of a stage during its forward, on the side stream (see _StageFn).
# Measured (same-box A/B): 53.2 vs 53.0 ms per step — the 42 five-microsecond transposes cost
# as much next to the forward GEMMs as between the backward ones; off by default.
PRETRANSPOSE_FILTERS = False


# ---- pooling AFTER res5.a's 1x1 projections ("projected pooling") --------------------------------
# The RoI head applies ROIAlign to the C4 feature map and then res5 (models/mask_rcnn_resnet.py:
# 168-176), whose first block reads the pooled map only through two 1x1 convolutions (conv1 and the
# shortcut conv4, each followed by AffineChannel2D).  ROIAlign is linear over positions per channel
# with no constant term, a bias-free 1x1 convolution is linear over channels per position: they
# commute exactly,
#     bn(conv1x1(roi_align(x))) == bn(roi_align(conv1x1(x))),
# so both projections run on the N*H*W map pixels (2 x 51 x 84 = 8 568 at the C2 shape) instead of
# the R*7*7 pooled ones (50 176), and ROIAlign pools their 512- / 2048-channel outputs with the
# affine (+ ReLU) in its epilogue.  Backward likewise: ROIAlign's adjoint first, then data and
# weight gradient of the projections on the map.  Same sums in a different order (fp32 rounding
# only); PROJECTED_POOLING = False restores the reference order.
PROJECTED_POOLING = _os.environ.get('MRCNN_PROJECTED_POOLING', '1') != '0'


class RoiSpec(object):
    """What ``building_block(..., roi=)`` needs to pool inside the stage node: ``rois`` (R, 5) float32
    device rows (batch, x1, y1, x2, y2), the pooled size ``outh`` x ``outw`` of the FULL bin grid,
    ``bin_stride`` (the stride of block a's 1x1 convolutions: only those bins are produced) and the
    optional processing ``order`` (functions.roi_align_2d.spatial_order)."""

    def __init__(self, rois, outh, outw, spatial_scale, bin_stride=1, order=None, sampling_ratio=0,
                 proj=None):
        # ``proj`` (inference only): the two projections of the map, ``projected_map(...)``, when the
        # caller pools several RoI sets from one map (MaskRCNN.predict_prepared: one head call per image)
        self.proj = proj
        self.rois = rois.contiguous()
        self.outh, self.outw = int(outh), int(outw)
        self.spatial_scale = float(spatial_scale)
        self.bin_stride = int(bin_stride)
        self.order = order
        self.sampling_ratio = int(sampling_ratio)

    @property
    def out_hw(self):
        bs = self.bin_stride
        return (self.outh + bs - 1) // bs, (self.outw + bs - 1) // bs


# The projections of one feature map, kept while the SAME map is pooled again without a graph
# (inference runs the head once per image and once per mask group on one batch's map): the map
# tensor itself is held so that its identity cannot be recycled; ``weights_changed`` drops the entry.
_proj_cache = {}


def projected_map(x, W1, W4):
    """(conv1x1(x, W1), conv1x1(x, W4)) of an NHWC map, bias-free and without epilogue — the inputs
    ``RoiSpec(proj=)`` takes.  Cached per (map, filters) while autograd is off."""
    x = nhwc(x)
    key = (id(x), x._version, x.data_ptr(), tuple(x.shape), id(W1), W1._version, W1.data_ptr(),
           id(W4), W4._version, W4.data_ptr())
    hit = _proj_cache.get('entry')
    if hit is not None and hit[0] == key:
        return hit[2], hit[3]
    d1 = make_desc(x.shape, W1.shape, 1, 0)
    d4 = make_desc(x.shape, W4.shape, 1, 0)
    z1 = _fwd_raw(x, nhwc(W1), d1, None, None, None, False)
    z4 = _fwd_raw(x, nhwc(W4), d4, None, None, None, False)
    _proj_cache['entry'] = (key, x, z1, z4)
    return z1, z4


def _roi_pool_affine(z, roi, scale, shift, relu):
    """relu?(roi_align(z) * scale + shift) for an NHWC map ``z`` (N, C, H, W logical)."""
    N, C, H, W = z.shape
    R = roi.rois.shape[0]
    oh, ow = roi.out_hw
    y = empty_nhwc((R, C, oh, ow), z.device)
    order = roi.order if R > 0 else None
    _lib.call('mrcnn_roi_align_fwd_affine', _lib.ptr(z), _lib.ptr(roi.rois), _lib.ptr(y), N, H, W, C, R,
              roi.outh, roi.outw, roi.bin_stride, roi.spatial_scale, roi.sampling_ratio,
              _lib.ptr(order) if order is not None else None, _lib.ptr(scale), _lib.ptr(shift),
              1 if relu else 0, _lib.stream_ptr())
    return y


def _roi_pool_bwd(g, roi, map_shape):
    """ROIAlign's adjoint: g (R, C, oh, ow) -> (N, C, H, W), pixel-owner form (no atomics)."""
    N, C, H, W = map_shape
    R = roi.rois.shape[0]
    gz = empty_nhwc((N, C, H, W), g.device)
    nbytes = _lib.load().mrcnn_roi_align_bwd_workspace_bytes(N, H, W, R, roi.outh, roi.outw, roi.bin_stride)
    ws = _lib.workspace(nbytes, g.device, 'roi_align_bwd')
    _lib.call('mrcnn_roi_align_bwd_ws', _lib.ptr(g), _lib.ptr(roi.rois), _lib.ptr(gz), N, H, W, C, R,
              roi.outh, roi.outw, roi.bin_stride, roi.spatial_scale, roi.sampling_ratio, _lib.ptr(ws),
              int(ws.numel() * ws.element_size()), _lib.stream_ptr())
    return gz



def _pool_bwd_alone(side, device):
    """The compute stream waits for the weight-gradient side stream before a pixel-owner ROIAlign
    backward inside a stage.  Round 6 finding (tests/test_gpu_model.py::test_training_is_bit_
    reproducible_run_to_run on the small test model, whose RoI head is small enough to use the side
    stream): with the weight-gradient GEMM of the SAME block running on the side stream — it reads the
    gradient tensor the ROIAlign backward reads, written by the data-gradient GEMM just before — a
    launch now and then lost ONE list entry's contribution in ONE component of 16 lanes (a handful
    of gx elements, different from run to run; captured and analysed on the host: inputs and tables
    identical, the sum short of exactly one term).  The same two kernels side by side in isolation
    (900 launches, the captured tensors included) never did; neither kernel uses scratch; four
    plain v_fma_f32 instead of the packed FMAs change nothing.  Two measures, either of which removes
    it on its own (each 8 of 8 runs clean): the pooled backward of a block is queued BEFORE that
    block's weight gradients go to the side stream (no second reader of its input is in flight), and
    the streams are serialised here.  Costs nothing at full size (the head's weight gradients do not
    use the side stream there), ~20 us in the small configurations that do.  Root cause not found."""
    if side is not None:
        torch.cuda.current_stream(device).wait_stream(side)


# set by building_block() around _StageFn.apply: inside forward() grad mode is always off and
# ctx.needs_input_grad reports the inputs' requires_grad flags even under torch.no_grad()
_STAGE_RECORDS_GRAPH = True


class _StageFn(torch.autograd.Function):

    @staticmethod
    def forward(ctx, x, strides, proj, poll, tail_rows, roi, *params):
        """``roi`` (a ``RoiSpec`` or None): with it ``x`` is the FEATURE MAP and block 0 (which must
        have a projection shortcut) runs its two 1x1 convolutions on the map and pools their outputs
        (projected pooling, see above) — the stage then equals ``stage(roi_align(x, roi))``.
        ``strides[i]`` / ``proj[i]``: conv1(/conv4) stride and has-projection flag of block
        i; ``poll``: optional callable invoked during backward at the stage's entry and after
        every block's weight gradients have been queued (parallel.DataParallelGradSync launches
        the gradient buckets that are complete); ``tail_rows``: None, or an int64 index tensor —
        the node then returns ``(average_pooling(y) (N,C,1,1), y[tail_rows])`` instead of the
        stage output y (the RoI head's two consumers of res5, models/mask_rcnn_resnet.py:186-195),
        and its backward forms the gradient entering the last block from both in ONE pass,
        ReLU mask included (``mrcnn_head_tail_bwd``); ``params``: per block
        W1,s1,b1,W2,s2,b2,W3,s3,b3 (+ W4,s4,b4 when proj[i])."""
        _lib.require_device(x, params[0])
        x = nhwc(x)
        blocks, saved, pos, wino_v = [], [x], 0, []
        h = x
        if roi is not None:
            if roi.proj is not None and _STAGE_RECORDS_GRAPH and any(ctx.needs_input_grad):
                raise ValueError('RoiSpec(proj=) is for graph-free calls (the projections would be '
                                 'outside the recorded graph)')
            if not (proj[0] and strides[0] == 1 and params[0].shape[2] == 1 and params[9].shape[2] == 1):
                raise ValueError('projected pooling needs a first block with 1x1 conv1 / conv4 at stride 1 '
                                 '(the RoI bins of the stride are selected by RoiSpec.bin_stride)')
        for bi, (stride, pj) in enumerate(zip(strides, proj)):
            n = 12 if pj else 9
            W1, s1, b1, W2, s2, b2, W3, s3, b3 = params[pos:pos + 9]
            W4, s4, b4 = params[pos + 9:pos + 12] if pj else (None, None, None)
            pooled_here = roi is not None and bi == 0
            d1 = make_desc(h.shape, W1.shape, stride, 0)
            if pooled_here:
                # conv1 on the map, then pooled with bn1 + ReLU in ROIAlign's epilogue
                z1 = roi.proj[0] if roi.proj is not None else _fwd_raw(h, nhwc(W1), d1, None, None, None, False)
                h1 = _roi_pool_affine(z1, roi, s1, b1, True)
                del z1
            else:
                h1 = _fwd_raw(h, nhwc(W1), d1, s1, b1, None, True)
            d2 = make_desc(h1.shape, W2.shape, 1, 1)
            v2 = None
            training = _STAGE_RECORDS_GRAPH and any(ctx.needs_input_grad)
            if uses_winograd(d2) and (WINOGRAD_TRAIN_FORWARD in (True, 'stage') or not training):
                # (with a weight gradient to come, the transformed input is kept for it)
                h2, v2 = wino_fwd(h1, nhwc(W2), d2, s2, b2, True,
                                  keep_v=training and bool(ctx.needs_input_grad[6 + pos + 3]),
                                  cache_for=None if training else W2,
                                  exact_signs=training and WINOGRAD_EXACT_SIGNS)
            else:
                h2 = _fwd_raw(h1, nhwc(W2), d2, s2, b2, None, True)
            d4 = None
            if pj:
                d4 = make_desc(h.shape, W4.shape, stride, 0)
                if pooled_here:
                    z4 = roi.proj[1] if roi.proj is not None else _fwd_raw(h, nhwc(W4), d4, None, None, None, False)
                    shortcut = _roi_pool_affine(z4, roi, s4, b4, False)
                    del z4
                else:
                    shortcut = _fwd_raw(h, nhwc(W4), d4, s4, b4, None, False)
            else:
                shortcut = h
            d3 = make_desc(h2.shape, W3.shape, 1, 0)
            y = _fwd_raw(h2, nhwc(W3), d3, s3, b3, shortcut, True)
            blocks.append(((d1, d2, d3, d4), (W1, W2, W3, W4), pos))
            if RELU_TAP is not None:
                RELU_TAP.append(('block', (h1, h2, y)))
            saved += [h1, h2, y, s1, s2, s3] + ([s4] if pj else [])
            wino_v.append(v2)
            pos += n
            h = y
        ctx.blocks = blocks
        ctx.proj = tuple(proj)
        ctx.poll = poll
        ctx.roi = roi
        ctx.n_saved = len(saved)
        ctx.wino_slots = [i for i, v in enumerate(wino_v) if v is not None]
        ctx.save_for_backward(*(saved + [wino_v[i] for i in ctx.wino_slots]))
        ctx.wT = None
        ctx.tail = tail_rows is not None
        if ctx.tail:
            # the two consumers of the stage output, produced here so that backward receives
            # their gradients separately (h itself is not an output of the node)
            R_, C_ = h.shape[0], h.shape[1]
            pooled = torch.empty((R_, C_), dtype=torch.float32, device=h.device)
            _lib.call('mrcnn_avgpool_fwd', _lib.ptr(h), _lib.ptr(pooled), R_, h.shape[2] * h.shape[3],
                      C_, _lib.stream_ptr())
            sub = h.permute(0, 2, 3, 1).index_select(0, tail_rows).permute(0, 3, 1, 2)
            slot = getattr(tail_rows, '_mrcnn_slot', None)     # built on the host by the caller
            if slot is None:
                slot = torch.full((R_,), -1, dtype=torch.int32, device=h.device)
                slot[tail_rows] = torch.arange(tail_rows.numel(), dtype=torch.int32, device=h.device)
            ctx.tail_slot = slot
        if PRETRANSPOSE_FILTERS and any(ctx.needs_input_grad):
            # The backward's forward-form dgrads need every filter flipped and transposed
            # (conv3 / conv4 with their affine scale folded in).  The weights are final for
            # this step, so the ~40 tiny transposes run now on the side stream, next to the
            # forward GEMMs, instead of between the backward GEMMs.
            dev = x.device
            side, main = wgrad_stream(dev), torch.cuda.current_stream(dev)
            side.wait_stream(main)          # the previous step's SGD update of the weights
            with torch.cuda.stream(side):
                ctx.wT = _stage_transposes(
                    blocks, [(params[p0 + 7], params[p0 + 10] if W4 is not None else None)
                             for _, (_, _, _, W4), p0 in blocks], dev)
                ctx.wT_ready = torch.cuda.Event()
                ctx.wT_ready.record(side)
        if ctx.tail:
            return pooled.reshape(R_, C_, 1, 1), sub
        return h

    @staticmethod
    def backward(ctx, gy, g_rows=None):
        saved = list(ctx.saved_tensors)
        wino_v = dict(zip(ctx.wino_slots, saved[ctx.n_saved:]))
        saved = saved[:ctx.n_saved]
        ng = ctx.needs_input_grad
        grads = [None] * len(ng)
        poll = ctx.poll
        if poll is not None:
            poll()                 # everything upstream of this stage has queued its gradients
        # unpack per-block activations
        acts, pos, xin = [], 1, saved[0]
        for pj in ctx.proj:
            n = 7 if pj else 6
            h1, h2, y, s1, s2, s3 = saved[pos:pos + 6]
            s4 = saved[pos + 6] if pj else None
            acts.append((xin, h1, h2, y, s1, s2, s3, s4))
            xin = y
            pos += n
        if not ctx.tail:
            gy = nhwc(gy)
        # small-M layers (backbone stages) leave CUs idle: their weight gradients go to a
        # second stream so that they can share the GPU with the next dgrad
        d_top = ctx.blocks[-1][0][2]
        dev_ = gy.device if gy is not None else g_rows.device
        side = wgrad_stream(dev_) if (
            SMALL_WGRAD_SIDE_STREAM and d_top.N * d_top.P * d_top.Q <= SMALL_WGRAD_MAX_PIXELS) else None
        if ctx.wT is not None:
            main = torch.cuda.current_stream(dev_)
            main.wait_event(ctx.wT_ready)
            for t in ctx.wT:
                for buf in t.values():
                    buf.record_stream(main)
            stage_wT = ctx.wT
        else:
            # all the stage's filter transposes in one launch
            stage_wT = _stage_transposes(ctx.blocks, [(a[6], a[7]) for a in acts], dev_)
        # gm: gradient w.r.t. the block output, already through that output's ReLU
        if ctx.tail:
            y_top = acts[-1][3]
            R_, C_, Hh, Ww = y_top.shape
            if gy is None:
                gy = torch.zeros((R_, C_), dtype=torch.float32, device=dev_)
            gp = gy.reshape(R_, C_).contiguous()
            gr = nhwc(g_rows) if g_rows is not None else None
            gm = empty_nhwc((R_, C_, Hh, Ww), dev_)
            _lib.call('mrcnn_head_tail_bwd', _lib.ptr(gp), _lib.ptr(gr),
                      _lib.ptr(ctx.tail_slot) if gr is not None else None, _lib.ptr(y_top),
                      _lib.ptr(gm), R_, Hh * Ww, C_, _lib.stream_ptr())
        else:
            gm = epilogue_bwd(gy, acts[-1][3], None)
        for i in range(len(acts) - 1, -1, -1):
            x, h1, h2, y, s1, s2, s3, s4 = acts[i]
            (d1, d2, d3, d4), (W1, W2, W3, W4), p0 = ctx.blocks[i]
            wT = stage_wT[i]
            base = 6 + p0                      # index of W1 among the forward inputs
            first = i == 0
            pooled_here = first and ctx.roi is not None
            # what the gradient leaving this block must be masked with: the previous block's
            # output ReLU (= this block's input); the stage input belongs to someone else
            xm = None if first else x
            gz4 = None
            if pooled_here and (ng[base + 9] or ng[0]):
                # the shortcut's gradient back on the map (ROIAlign's adjoint commutes with the
                # per-channel scale s4, which stays folded into conv4's filter / gradient rows).
                # Queued BEFORE this block's weight gradients go to the side stream, and with the side
                # stream drained: see _pool_bwd_alone.
                _pool_bwd_alone(side, dev_)
                gz4 = _roi_pool_bwd(gm, ctx.roi, (x.shape[0], d4.K, x.shape[2], x.shape[3]))
            if ng[base + 6]:
                grads[base + 6] = _wgrad_raw(d3, h2, gm, W3, None, None, side, row_scale=s3)
            if pooled_here:
                if ng[base + 9]:
                    grads[base + 9] = _wgrad_raw(d4, x, gz4, W4, None, None, side, row_scale=s4)
            elif W4 is not None and ng[base + 9]:
                grads[base + 9] = _wgrad_raw(d4, x, gm, W4, None, None, side, row_scale=s4)
            gh2 = _dgrad_raw(d3, gm, nhwc(W3), None, None, fold_scale=s3,
                             out_mask_y=h2, out_scale=s2, wT=wT.get('3'))
            if uses_winograd(d2):
                if ng[base + 3]:
                    if WINOGRAD_WGRAD:
                        v2 = wino_v.get(i)
                        grads[base + 3] = _wino_wgrad(d2, h1 if v2 is None else None, v2, gh2, W2)
                    else:
                        grads[base + 3] = _wgrad_raw(d2, h1, gh2, W2, None, None, side)
                if WINOGRAD_DGRAD:
                    gh1 = wino_dgrad(d2, gh2, nhwc(W2), out_scale=s1, out_mask_y=h1)
                else:
                    gh1 = _dgrad_raw(d2, gh2, nhwc(W2), None, None, out_mask_y=h1, out_scale=s1)
            else:
                if ng[base + 3]:
                    grads[base + 3] = _wgrad_raw(d2, h1, gh2, W2, None, None, side)
                gh1 = _dgrad_raw(d2, gh2, nhwc(W2), None, None, out_mask_y=h1, out_scale=s1,
                                 wT=wT.get('2'))
            if pooled_here and (ng[base] or ng[0]):
                _pool_bwd_alone(side, dev_)
                gh1 = _roi_pool_bwd(gh1, ctx.roi, (x.shape[0], d1.K, x.shape[2], x.shape[3]))
            if ng[base]:
                grads[base] = _wgrad_raw(d1, x, gh1, W1, None, None, side)
            if poll is not None:
                poll()             # this block's weight gradients are queued
            if first and not ng[0]:
                break
            if pooled_here:
                gx = _dgrad_raw(d1, gh1, nhwc(W1), None, None, wT=wT.get('1'))
                gm = _dgrad_raw(d4, gz4, nhwc(W4), None, None, fold_scale=s4, out=gx, accum=True,
                                wT=wT.get('4'))
                continue
            if W4 is None:
                gm = _dgrad_raw(d1, gh1, nhwc(W1), None, None, res_g=gm, out_mask_y=xm,
                                wT=wT.get('1'))
            elif d1.stride == 1:
                gx = _dgrad_raw(d1, gh1, nhwc(W1), None, None, wT=wT.get('1'))
                gm = _dgrad_raw(d4, gm, nhwc(W4), None, None, fold_scale=s4, out=gx, accum=True,
                                out_mask_y=xm, wT=wT.get('4'))
            else:
                gx = _dgrad_raw(d1, gh1, nhwc(W1), None, None, wT=wT.get('1'))
                gm = _dgrad_raw(d4, gm, nhwc(W4), None, None, fold_scale=s4, out=gx, accum=True,
                                wT=wT.get('4'))
                if xm is not None:
                    gm = epilogue_bwd(gm, xm, None)
        if ng[0]:
            grads[0] = gm
        return tuple(grads)


def building_block(x, blocks, first_stride=None, poll=None, tail_rows=None, roi=None):
    """A chain of Bottleneck links (chainer BuildingBlock) as one fused autograd node.  With
    ``tail_rows`` (int64 row indices) it returns ``(average_pooling_2d(y, map size), y[tail_rows])``
    instead of y — see _StageFn.forward.  With ``roi`` (a ``RoiSpec``) ``x`` is the feature map and
    the result is ``building_block(roi_align(x, roi), ...)`` (projected pooling)."""
    strides, proj, params = [], [], []
    for i, b in enumerate(blocks):
        strides.append(b.conv1.stride if not (i == 0 and first_stride is not None) else first_stride)
        proj.append(bool(b.projection))
        params += [b.conv1.W, b.bn1.W, b.bn1.b, b.conv2.W, b.bn2.W, b.bn2.b,
                   b.conv3.W, b.bn3.W, b.bn3.b]
        if b.projection:
            params += [b.conv4.W, b.bn4.W, b.bn4.b]
    global _STAGE_RECORDS_GRAPH
    _STAGE_RECORDS_GRAPH = torch.is_grad_enabled()
    try:
        return _StageFn.apply(x, tuple(strides), tuple(proj), poll, tail_rows, roi, *params)
    finally:
        _STAGE_RECORDS_GRAPH = True
