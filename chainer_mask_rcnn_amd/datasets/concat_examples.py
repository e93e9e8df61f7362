"""concat_examples — /root/reference/chainer_mask_rcnn/datasets/concat_examples.py:6-34
(chainer.dataset.convert._concat_arrays / to_device underneath).

Same arguments.  ``device`` is a torch device (or None: leave everything where it is);
entries listed in ``indices_to_device`` become device tensors, the others stay NumPy.
Image entries that are already device tensors (``MaskRCNNTransform``) are zero-padded and
stacked on the device, in channels-last memory — the layout the extractor reads."""
import numpy as np
import torch


def _pad_shape(arrays):
    shape = np.array(arrays[0].shape, dtype=int)
    for a in arrays[1:]:
        if np.any(shape != a.shape):
            np.maximum(shape, a.shape, shape)
    return tuple(int(s) for s in shape)


def _concat_arrays(arrays, padding):
    first = arrays[0]
    if isinstance(first, torch.Tensor):
        if padding is None:
            return torch.stack(list(arrays))
        shape = (len(arrays),) + _pad_shape(arrays)
        fmt = torch.channels_last if len(shape) == 4 else torch.contiguous_format
        out = torch.full(shape, padding, dtype=first.dtype, device=first.device).contiguous(
            memory_format=fmt)
        for i, a in enumerate(arrays):
            out[(i,) + tuple(slice(0, d) for d in a.shape)] = a
        return out
    if not isinstance(first, np.ndarray):
        arrays = [np.asarray(a) for a in arrays]
        first = arrays[0]
    if padding is None:
        return np.concatenate([a[None] for a in arrays])
    shape = (len(arrays),) + _pad_shape(arrays)
    if all(a.shape == shape[1:] for a in arrays):
        # nothing to pad (the usual case for a batch of equally sized images): one pass, no fill
        out = np.empty(shape, dtype=first.dtype)
        for i, a in enumerate(arrays):
            out[i] = a
        return out
    out = np.full(shape, padding, dtype=first.dtype)
    for i, a in enumerate(arrays):
        out[(i,) + tuple(slice(0, d) for d in a.shape)] = a
    return out


def _to_device(device, x):
    if device is None:
        return x
    if isinstance(x, torch.Tensor):
        return x.to(device)
    return torch.as_tensor(np.ascontiguousarray(x)).to(device)


def concat_examples(batch, device=None, padding=None,
                    indices_concat=None, indices_to_device=None):
    """Column-wise collation of a list of example tuples.  Field ``i`` is stacked (zero- or
    ``padding[i]``-padded to the largest shape) when ``i`` is in ``indices_concat`` and moved
    to ``device`` when ``i`` is in ``indices_to_device``; both default to every field."""
    if not batch:
        raise ValueError('batch is empty')
    n_fields = len(batch[0])
    stacked = set(range(n_fields) if indices_concat is None else indices_concat)
    moved = set(range(n_fields) if indices_to_device is None else indices_to_device)
    pads = padding if isinstance(padding, tuple) else (padding,) * n_fields

    def collate(i, column):
        if i in stacked:
            column = _concat_arrays(column, pads[i])
            return _to_device(device, column) if i in moved else column
        return [_to_device(device, c) for c in column] if i in moved else column

    return tuple(collate(i, [ex[i] for ex in batch]) for i in range(n_fields))
