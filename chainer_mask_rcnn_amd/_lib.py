"""ctypes binding of libmrcnn_hip.so (include/mrcnn_hip.h).

The HIP library is the product: if it cannot be loaded, every op fails loudly —
there is no CPU or eager-PyTorch fallback anywhere in this package.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('MRCNN_HIP_LIB') or os.path.join(_HERE, 'libmrcnn_hip.so')   # override: developer A/B builds

EPI_BIAS, EPI_AFFINE, EPI_RESIDUAL, EPI_RELU, EPI_ACCUM, EPI_EXACT_SIGNS = 1, 2, 4, 8, 16, 32

c_int, c_i64, c_f32, c_vp = ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p


class ConvDesc(ctypes.Structure):
    """mrcnn_conv_desc."""
    _fields_ = [(k, c_int) for k in
                ('N', 'H', 'W', 'C', 'K', 'R', 'S', 'stride', 'pad', 'P', 'Q')]


_DP = ctypes.POINTER(ConvDesc)


# name -> (restype, argtypes); every symbol declared in include/mrcnn_hip.h
SIGNATURES = {
    'mrcnn_last_error': (ctypes.c_char_p, []),
    'mrcnn_abi_version': (c_int, []),
    'mrcnn_device_info': (c_int, [ctypes.POINTER(c_int), ctypes.c_char_p, c_int]),
    'mrcnn_profile_enable': (c_int, [c_int]),
    'mrcnn_profile_num_kinds': (c_int, []),
    'mrcnn_profile_kind_name': (ctypes.c_char_p, [c_int]),
    'mrcnn_profile_summary': (c_int, [c_int, ctypes.POINTER(ctypes.c_double),
                                      ctypes.POINTER(ctypes.c_double),
                                      ctypes.POINTER(ctypes.c_double),
                                      ctypes.POINTER(c_i64)]),
    'mrcnn_roi_align_fwd': (c_int, [c_vp, c_vp, c_vp] + [c_int] * 7 + [c_f32, c_int, c_vp]),
    'mrcnn_roi_align_bwd': (c_int, [c_vp, c_vp, c_vp] + [c_int] * 7 + [c_f32, c_int, c_vp]),
    'mrcnn_roi_align_fwd_ex': (c_int, [c_vp, c_vp, c_vp] + [c_int] * 8 + [c_f32, c_int, c_vp, c_vp]),
    'mrcnn_roi_align_fwd_affine': (c_int, [c_vp, c_vp, c_vp] + [c_int] * 8 + [c_f32, c_int, c_vp, c_vp, c_vp,
                                           c_int, c_vp]),
    'mrcnn_roi_align_bwd_workspace_bytes': (c_i64, [c_int] * 7),
    'mrcnn_sparse3x3_gather': (c_int, [c_vp, c_vp, c_vp] + [c_int] * 6 + [c_vp, c_vp, c_vp]),
    'mrcnn_sparse3x3_scatter': (c_int, [c_vp, c_vp] + [c_int] * 4 + [c_vp, c_vp]),
    'mrcnn_roi_align_bwd_ex': (c_int, [c_vp, c_vp, c_vp] + [c_int] * 8 + [c_f32, c_int, c_vp, c_vp]),
    'mrcnn_roi_align_bwd_ws': (c_int, [c_vp, c_vp, c_vp] + [c_int] * 8 + [c_f32, c_int, c_vp, c_i64, c_vp]),
    'mrcnn_affine_fwd': (c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_int, c_vp]),
    'mrcnn_colsum_workspace_bytes': (c_i64, [c_int]),
    'mrcnn_affine_bwd': (c_int, [c_vp] * 6 + [c_i64, c_int, c_vp, c_vp]),
    'mrcnn_decode_clip': (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_f32, c_f32, c_f32, c_vp]),
    'mrcnn_topk_workspace_bytes': (c_i64, [c_int]),
    'mrcnn_topk_desc': (c_int, [c_vp, c_vp, c_int, c_int, c_vp, c_vp, c_vp, c_vp]),
    'mrcnn_topk_desc_batched': (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp]),
    'mrcnn_detect_sort_workspace_bytes': (c_i64, [c_int, c_int]),
    'mrcnn_detect_sort': (c_int, [c_vp, c_vp, c_int, c_int, c_f32, c_vp, c_vp, c_vp, c_vp, c_vp]),
    'mrcnn_detect_compact': (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp]),
    'mrcnn_gather_rows': (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_vp, c_vp]),
    'mrcnn_nms_workspace_bytes': (c_i64, [c_int, c_int]),
    'mrcnn_nms_sorted': (c_int, [c_vp, c_vp, c_int, c_f32, c_int, c_vp, c_vp, c_vp, c_vp]),
    'mrcnn_nms_sorted_batched': (c_int, [c_vp, c_vp, c_int, c_int, c_f32, c_int, c_vp, c_vp,
                                         c_vp, c_vp]),
    'mrcnn_conv2d_split_workspace_bytes': (c_i64, []),
    'mrcnn_set_tuning': (c_int, [ctypes.c_char_p, c_int]),
    'mrcnn_conv2d_fwd': (c_int, [_DP] + [c_vp] * 7 + [c_int, c_vp, c_vp]),
    'mrcnn_conv2d_dgrad': (c_int, [_DP, c_vp, c_vp, c_vp, c_int, c_vp]),
    'mrcnn_conv2d_wgrad_workspace_bytes': (c_i64, [_DP]),
    'mrcnn_conv2d_wgrad': (c_int, [_DP, c_vp, c_vp, c_vp, c_vp, c_vp]),
    'mrcnn_conv2d_dgrad_ex': (c_int, [_DP, c_vp, c_vp, c_vp, c_int] + [c_vp] * 8),
    'mrcnn_conv2d_wgrad_ex': (c_int, [_DP] + [c_vp] * 8),
    'mrcnn_conv3x3_wino_v_bytes': (c_i64, [_DP]),
    'mrcnn_conv3x3_wino_workspace_bytes': (c_i64, [_DP]),
    'mrcnn_conv3x3_wino_u_bytes': (c_i64, [_DP]),
    'mrcnn_conv3x3_wino_filter': (c_int, [_DP] + [c_vp] * 3),
    'mrcnn_conv3x3_wino_fwd': (c_int, [_DP] + [c_vp] * 6 + [c_int] + [c_vp] * 3),
    'mrcnn_conv3x3_wino_fixup_count': (c_int, [_DP, c_vp, c_vp, c_vp]),
    'mrcnn_conv3x3_wino_dgrad': (c_int, [_DP] + [c_vp] * 8),
    'mrcnn_conv3x3_wino_wgrad': (c_int, [_DP] + [c_vp] * 7),
    'mrcnn_filter_flip_transpose': (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp, c_vp]),
    'mrcnn_filter_flip_transpose_batched': (c_int, [c_int] + [c_vp] * 8),
    'mrcnn_conv2d_dgrad_wt': (c_int, [_DP, c_vp, c_vp, c_vp, c_int] + [c_vp] * 8),
    'mrcnn_conv_stem_fwd': (c_int, [c_vp] * 6 + [c_int] * 5 + [c_vp]),
    'mrcnn_deconv2x2s2_fwd': (c_int, [c_vp] * 4 + [c_int] * 6 + [c_vp]),
    'mrcnn_deconv2x2s2_fwd_wt': (c_int, [c_vp] * 4 + [c_int] * 6 + [c_vp]),
    'mrcnn_deconv2x2s2_dgrad': (c_int, [c_vp] * 3 + [c_int] * 5 + [c_vp]),
    'mrcnn_deconv2x2s2_wgrad_workspace_bytes': (c_i64, [c_int] * 5),
    'mrcnn_deconv2x2s2_wgrad': (c_int, [c_vp] * 3 + [c_int] * 5 + [c_vp, c_vp]),
    'mrcnn_epilogue_bwd': (c_int, [c_vp] * 4 + [c_i64, c_int, c_vp]),
    'mrcnn_colsum': (c_int, [c_vp, c_vp, c_i64, c_int, c_vp, c_vp]),
    'mrcnn_maxpool3x3s2p1_fwd': (c_int, [c_vp, c_vp] + [c_int] * 6 + [c_vp]),
    'mrcnn_avgpool_fwd': (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_vp]),
    'mrcnn_avgpool_bwd': (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp]),
    'mrcnn_head_tail_bwd': (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp]),
    'mrcnn_loss_workspace_bytes': (c_i64, [c_int]),
    'mrcnn_sigmoid_ce': (c_int, [c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp]),
    'mrcnn_mask_sigmoid_ce': (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp, c_vp, c_vp,
                                      c_vp]),
    'mrcnn_softmax_ce': (c_int, [c_vp, c_int, c_vp, c_int, c_int, c_vp, c_vp, c_int, c_vp,
                                 c_vp]),
    'mrcnn_smooth_l1': (c_int, [c_vp, c_int, c_vp, c_vp, c_vp, c_int, c_f32, c_vp, c_vp, c_vp,
                                c_vp]),
    'mrcnn_softmax': (c_int, [c_vp, c_int, c_vp, c_int, c_int, c_int, c_vp]),
    'mrcnn_sgd_momentum_wd': (c_int, [c_vp, c_vp, c_vp, c_i64, c_f32, c_f32, c_f32, c_f32,
                                      c_vp]),
    'mrcnn_sgd_momentum_wd_ex': (c_int, [c_vp, c_vp, c_vp, c_i64, c_f32, c_f32, c_f32, c_f32,
                                         c_int, c_vp]),
    'mrcnn_bbox_iou_argmax': (c_int, [c_vp, c_int, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp]),
    'mrcnn_anchor_labels': (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_f32, c_f32, c_vp, c_vp]),
    'mrcnn_anchor_targets_finish': (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_int,
                                            c_int, c_vp, c_vp, c_vp]),
    'mrcnn_proposal_targets_gather': (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int,
                                              ctypes.POINTER(c_f32), ctypes.POINTER(c_f32),
                                              c_vp, c_vp, c_vp, c_vp, c_vp]),
    'mrcnn_mask_targets': (c_int, [c_vp, c_int, c_int, c_int, c_vp, c_vp, c_int, c_int, c_int,
                                   c_vp, c_vp]),
    'mrcnn_allreduce_unique_id': (c_int, [c_vp]),
    'mrcnn_allreduce_init': (c_int, [c_vp, c_int, c_int, ctypes.POINTER(c_vp)]),
    'mrcnn_allreduce_destroy': (c_int, [c_vp]),
    'mrcnn_allreduce_info': (c_int, [c_vp, ctypes.POINTER(c_int), ctypes.POINTER(c_int),
                                     ctypes.POINTER(c_int), ctypes.c_char_p, c_int]),
    'mrcnn_allreduce_bucket': (c_int, [c_vp, c_vp, c_i64, c_int, c_vp, c_vp]),
    'mrcnn_allreduce_wait': (c_int, [c_vp, c_vp]),
    'mrcnn_allreduce_broadcast': (c_int, [c_vp, c_vp, c_i64, c_int, c_vp]),
    'mrcnn_allreduce_timing': (c_int, [c_vp, c_int]),
    'mrcnn_allreduce_bucket_times': (c_int, [c_vp, c_int, ctypes.POINTER(ctypes.c_double),
                                             ctypes.POINTER(ctypes.c_double),
                                             ctypes.POINTER(c_i64)]),
    'mrcnn_prepare_image': (c_int, [c_vp, c_int, c_int, c_int, c_int, ctypes.c_double,
                                    ctypes.POINTER(c_f32), c_vp, c_int, c_int, c_int, c_int, c_int,
                                    c_int, c_vp]),
    'mrcnn_paste_masks': (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp]),
    'mrcnn_decode_cls_boxes': (c_int, [c_vp, c_vp, c_int, c_vp, c_int, c_int, c_f32,
                                       ctypes.POINTER(ctypes.c_double),
                                       ctypes.POINTER(ctypes.c_double), c_f32, c_f32, c_vp]),
}

_lib = None


class MrcnnHipError(RuntimeError):
    pass


_load_error = None   # a failed MRCNN_TUNE application: load() keeps raising it (see load())


def load():
    """Load libmrcnn_hip.so (built by ``__graft_entry__.build()``)."""
    global _lib, _load_error
    if _lib is not None:
        return _lib
    if _load_error is not None:
        raise _load_error
    if not os.path.exists(LIB_PATH):
        raise MrcnnHipError(
            'libmrcnn_hip.so not found at %s — build it with '
            '`python -c "import __graft_entry__ as g; g.build()"` or '
            '`make -C chainer_mask_rcnn_amd/csrc`. There is no fallback path.' % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    # MRCNN_TUNE="name=value,...": mrcnn_set_tuning knobs applied when the library is loaded
    # (e.g. MRCNN_TUNE=split_bf16=0 selects the fp32-MFMA GEMM kernels).  Every entry is parsed
    # before any is applied; a knob the library rejects leaves the process-wide library state
    # half-tuned, so the error is remembered and every later load() raises it again.
    knobs = []
    for kv in os.environ.get('MRCNN_TUNE', '').split(','):
        if not kv.strip():
            continue
        k, sep, v = kv.partition('=')
        try:
            value = int(v)
        except ValueError:
            sep = ''
        if not sep or not k.strip():
            _load_error = MrcnnHipError(
                "MRCNN_TUNE: expected 'name=integer[,name=integer...]', got %r" % kv)
            raise _load_error
        knobs.append((k.strip(), value))
    for k, value in knobs:
        rc = lib.mrcnn_set_tuning(k.encode(), int(value))
        if rc != 0:
            msg = lib.mrcnn_last_error()
            _load_error = MrcnnHipError('MRCNN_TUNE: set_tuning(%s=%d) failed (rc=%d): %s; knobs '
                                        'applied before it stay set — fix the variable and restart'
                                        % (k, value, rc, msg.decode() if msg else ''))
            raise _load_error
    _lib = lib
    return lib


def set_tuning(name, value):
    """mrcnn_set_tuning(name, value): kernel-selection knobs of libmrcnn_hip.so (include/mrcnn_hip.h)."""
    check(load().mrcnn_set_tuning(name.encode(), int(value)), 'set_tuning(%s)' % name)


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr())


def check(rc, what=''):
    if rc != 0:
        msg = load().mrcnn_last_error()
        raise MrcnnHipError('%s failed (rc=%d): %s' % (what, rc, msg.decode() if msg else ''))


def call(name, *args):
    """Call an int-returning entry point on the current stream; raise on error."""
    lib = load()
    rc = getattr(lib, name)(*args)
    check(rc, name)


def require_device(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise MrcnnHipError(
                'chainer_mask_rcnn_amd ops run only on a ROCm device tensor '
                '(got a %s tensor); there is no CPU path.' % t.device)


_ws_cache = {}


def workspace(nbytes, device, tag='default'):
    """Caller-owned scratch, cached per (device, current stream, tag); grows monotonically.
    Scratch is stream-ordered: two streams that run the same op concurrently (the weight-gradient
    side stream, the deferred-gradient stream, the input pipeline's stream, the frozen-prefix
    prefetch) must never share a buffer, so the stream is part of the key."""
    dev = torch.device(device)
    sid = torch.cuda.current_stream(dev).cuda_stream if dev.type == 'cuda' else 0
    key = (str(device), sid, tag)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf
