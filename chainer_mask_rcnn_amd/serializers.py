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
    data = np.load(path)
    fused = _fused_parts(model)
    used = set()
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
