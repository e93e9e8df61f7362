"""Chainer-compatible `.npz` snapshots of the model.

The reference saves `snapshot_object(model.mask_rcnn, 'snapshot_model.npz')`
(/root/reference/examples/train_common.py:276-283) and loads with
`chainer.serializers.load_npz(pretrained_model, self)` (models/mask_rcnn_resnet.py:115-116).
Keys are chainer link paths — `extractor/res4/b5/conv3/W`, `extractor/bn1/b`, `rpn/loc/W`,
`head/cls_loc/b`, `head/deconv6/W` … — with chainer's shapes: conv `(out,in,kh,kw)`,
deconv `(in,out,kh,kw)`, linear `(out,in)`, affine `(C,)`; the same set of names the reference's
`examples/coco/convert_caffe2_to_chainer.py:47-249` fills.  The fused RPN / head filters of this
build are split into (and assembled from) their reference-named parts.
"""
import numpy as np
import torch

# fused parameter -> [(reference link name, row range)]
def _fused_parts(model):
    A = model.rpn.n_anchor
    n_class = model.head.n_class
    return {
        'rpn.loc_score': [('rpn/loc', 0, 4 * A), ('rpn/score', 4 * A, 5 * A)],
        'head.cls_loc_score': [('head/cls_loc', 0, 4 * n_class),
                               ('head/score', 4 * n_class, 5 * n_class)],
    }


def state_arrays(model):
    """{chainer key: ndarray} for every parameter of a MaskRCNNResNet."""
    from . import optimizers
    optimizers.flush_all()                 # deferred parameter updates, if any
    fused = _fused_parts(model)
    out = {}
    for name, p in model.named_parameters():
        arr = p.detach().cpu().contiguous().numpy()
        owner, leaf = name.rsplit('.', 1)
        if owner in fused:
            for ref, lo, hi in fused[owner]:
                out['%s/%s' % (ref, leaf)] = arr[lo:hi].copy()
        else:
            out[name.replace('.', '/')] = arr
    return out


def save_npz(path, model):
    np.savez(path, **state_arrays(model))


def load_npz(path, model, strict=True):
    """Copy a chainer `.npz` snapshot into `model` (shapes must match, as `np.copyto` in the
    reference's converter would require)."""
    from . import optimizers
    from .functions import conv
    # a held-back weight gradient + SGD slice of the previous step must land BEFORE the snapshot
    # is copied in (it would otherwise update the freshly loaded head weights)
    optimizers.flush_all()
    data = np.load(path)
    fused = _fused_parts(model)
    used = set()
    conv.weights_changed()                 # cached transformed filters of inference calls
    ext = getattr(model, 'extractor', None)
    if ext is not None and hasattr(ext, 'drop_prefetched'):
        ext.drop_prefetched()              # a frozen prefix computed with the old weights
    with torch.no_grad():
        for name, p in model.named_parameters():
            owner, leaf = name.rsplit('.', 1)
            if owner in fused:
                for ref, lo, hi in fused[owner]:
                    key = '%s/%s' % (ref, leaf)
                    if key not in data:
                        if strict:
                            raise KeyError(key)
                        continue
                    a = data[key]
                    if tuple(a.shape) != tuple(p[lo:hi].shape):
                        raise ValueError('%s: snapshot %s vs model %s'
                                         % (key, a.shape, tuple(p[lo:hi].shape)))
                    p[lo:hi].copy_(torch.from_numpy(np.ascontiguousarray(a)))
                    used.add(key)
                continue
            key = name.replace('.', '/')
            if key not in data:
                if strict:
                    raise KeyError(key)
                continue
            a = data[key]
            if tuple(a.shape) != tuple(p.shape):
                raise ValueError('%s: snapshot %s vs model %s' % (key, a.shape, tuple(p.shape)))
            p.copy_(torch.from_numpy(np.ascontiguousarray(a)))
            used.add(key)
    if strict:
        extra = [k for k in data.files if k not in used]
        if extra:
            raise KeyError('snapshot has keys the model does not: %s' % extra[:5])
    return model


# ---------------------------------------------------------------------------------------------
# Detectron (Caffe2) -> model: the mapping of the reference's
# /root/reference/examples/coco/convert_caffe2_to_chainer.py:46-249, as a rule instead of 200
# literal lines (so it also covers ResNet-101's 23 res4 blocks):
#   conv1_w[:, ::-1]                    BGR -> RGB input channels                      (:47)
#   res{S}_{i}_branch2{a,b,c}_w         -> res{S}/{a|b{i}}/conv{1,2,3}/W               (:52-216)
#   res{S}_{i}_branch1_w                -> res{S}/a/conv4/W   (projection shortcut)
#   ..._bn_s / ..._bn_b                 -> the AffineChannel2D W / b that follows
#   rpn_bbox_pred_{w,b}                 (A,4,..)[:, [1,0,3,2]]   (dx,dy,dw,dh)->(dy,dx,dh,dw)  (:186-195)
#   bbox_pred_{w,b}                     (n_class,4,..)[:, [1,0,3,2]]                   (:233-243)
#   mask_fcn_logits_{w,b}[1:]           background mask channel dropped                (:248-249)
# res2..res4 live in `extractor`, res5 in `head`.  Momentum blobs, fc1000 and the (zero) conv
# biases of the BN-folded branches are ignored, as the reference does (:255-262).
# ---------------------------------------------------------------------------------------------
DETECTRON_MEAN = (122.7717, 115.9465, 102.9801)      # params.yaml written by the converter (:296)
_BLOCKS = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3)}


def detectron_to_chainer(blobs, n_layers=50, n_anchor=15, n_class=81):
    """{Detectron blob name: ndarray} -> {chainer key: ndarray} (reference key names / shapes)."""
    if n_layers not in _BLOCKS:
        raise ValueError('n_layers must be 50 or 101')
    out = {}

    def put(key, arr):
        out[key] = np.ascontiguousarray(arr, dtype=np.float32)

    put('extractor/conv1/W', np.asarray(blobs['conv1_w'])[:, ::-1])
    put('extractor/conv1/b', blobs['conv1_b'])
    put('extractor/bn1/W', blobs['res_conv1_bn_s'])
    put('extractor/bn1/b', blobs['res_conv1_bn_b'])
    branches = (('branch2a', 1), ('branch2b', 2), ('branch2c', 3))
    for stage, n_blocks in zip((2, 3, 4, 5), _BLOCKS[n_layers]):
        owner = 'head' if stage == 5 else 'extractor'
        for i in range(n_blocks):
            block = '%s/res%d/%s' % (owner, stage, 'a' if i == 0 else 'b%d' % i)
            src = 'res%d_%d_' % (stage, i)
            pairs = branches + ((('branch1', 4),) if i == 0 else ())
            for br, k in pairs:
                put('%s/conv%d/W' % (block, k), blobs[src + br + '_w'])
                put('%s/bn%d/W' % (block, k), blobs[src + br + '_bn_s'])
                put('%s/bn%d/b' % (block, k), blobs[src + br + '_bn_b'])
    put('rpn/conv1/W', blobs['conv_rpn_w'])
    put('rpn/conv1/b', blobs['conv_rpn_b'])
    perm = [1, 0, 3, 2]                        # (dx, dy, dw, dh) -> (dy, dx, dh, dw)
    W = np.asarray(blobs['rpn_bbox_pred_w'])
    put('rpn/loc/W', W.reshape((n_anchor, 4) + W.shape[1:])[:, perm].reshape(W.shape))
    put('rpn/loc/b', np.asarray(blobs['rpn_bbox_pred_b']).reshape(n_anchor, 4)[:, perm].reshape(-1))
    put('rpn/score/W', blobs['rpn_cls_logits_w'])
    put('rpn/score/b', blobs['rpn_cls_logits_b'])
    put('head/score/W', blobs['cls_score_w'])
    put('head/score/b', blobs['cls_score_b'])
    W = np.asarray(blobs['bbox_pred_w'])
    put('head/cls_loc/W', W.reshape(n_class, 4, -1)[:, perm].reshape(W.shape))
    put('head/cls_loc/b', np.asarray(blobs['bbox_pred_b']).reshape(n_class, 4)[:, perm].reshape(-1))
    put('head/deconv6/W', blobs['conv5_mask_w'])
    put('head/deconv6/b', blobs['conv5_mask_b'])
    put('head/mask/W', np.asarray(blobs['mask_fcn_logits_w'])[1:])   # remove background class
    put('head/mask/b', np.asarray(blobs['mask_fcn_logits_b'])[1:])
    return out


def load_detectron_pkl(path):
    """`model_final.pkl` of a Detectron run -> its blob dict (:33-38)."""
    import pickle
    with open(path, 'rb') as f:
        return pickle.load(f, encoding='latin-1')['blobs']


def load_arrays(arrays, model, strict=True):
    """Copy a {chainer key: ndarray} dict into `model` (same rules as ``load_npz``)."""
    import io
    buf = io.BytesIO()
    np.savez(buf, **arrays)
    buf.seek(0)
    return load_npz(buf, model, strict=strict)


def load_detectron(blobs_or_path, model, n_layers=None):
    """Fill a MaskRCNNResNet from Detectron weights (blob dict or `.pkl` path): what running
    the reference converter and then `load_npz` on its output does.  Use
    ``mean=serializers.DETECTRON_MEAN`` for such a model."""
    blobs = load_detectron_pkl(blobs_or_path) if isinstance(blobs_or_path, str) else blobs_or_path
    if n_layers is None:
        n_layers = 101 if len(model.extractor.res4._names) == 23 else 50
    arrays = detectron_to_chainer(blobs, n_layers, model.rpn.n_anchor, model.head.n_class)
    return load_arrays(arrays, model)
