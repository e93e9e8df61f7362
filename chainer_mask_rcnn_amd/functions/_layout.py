"""Layout helpers: tensors keep the reference's logical NCHW shape and are stored
channels-last (NHWC), the layout every HIP kernel of this package reads."""
import torch


def nhwc(x):
    """Return x (logical NCHW) backed by dense NHWC memory (no copy if it already is)."""
    assert x.dim() == 4
    n, c, h, w = x.shape
    want = (h * w * c, 1, w * c, c)
    if all(x.shape[i] == 1 or x.stride(i) == want[i] for i in range(4)):
        return x
    out = torch.empty((n, h, w, c), dtype=x.dtype, device=x.device).permute(0, 3, 1, 2)
    out.copy_(x)
    return out


def empty_nhwc(shape, device, dtype=torch.float32):
    n, c, h, w = shape
    return torch.empty((n, h, w, c), dtype=dtype, device=device).permute(0, 3, 1, 2)


def zeros_nhwc(shape, device, dtype=torch.float32):
    n, c, h, w = shape
    return torch.zeros((n, h, w, c), dtype=dtype, device=device).permute(0, 3, 1, 2)
