"""MomentumSGD + WeightDecay as the reference configures them
(/root/reference/examples/train_common.py:176-190; chainer rules):

    g += rate * p ;  v = momentum * v - lr * g ;  p += v      for every enabled parameter

All enabled parameters live in ONE flat fp32 arena (values, gradients and momenta), so
the update is a single HIP launch and the data-parallel gradient exchange is a handful
of large contiguous RCCL all-reduces instead of one message per parameter.  Weight
gradients are written by the wgrad kernels directly into the gradient arena.
"""
import torch

from . import _lib


class WeightDecay(object):
    """chainer.optimizer.WeightDecay(rate) hook."""

    def __init__(self, rate):
        self.rate = rate


def disable_update(module):
    """``link.disable_update()``: the module's parameters are never updated.  Their
    gradients are not computed either (the reference computes and discards them)."""
    for p in module.parameters():
        p.requires_grad_(False)


class ParamArena(object):
    """Flat storage for a list of dense parameters (16-byte aligned slices), ordered so
    that gradients become ready front-to-back during backward (reverse registration)."""

    ALIGN = 4  # floats

    def __init__(self, params):
        self.params = list(params)
        dev = self.params[0].device
        offs, total = [], 0
        for p in self.params:
            if not _is_dense(p):
                raise ValueError('arena parameters must be dense')
            offs.append(total)
            total += (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.offsets, self.size = offs, total
        self.values = torch.zeros(total, dtype=torch.float32, device=dev)
        self.grads = torch.zeros(total, dtype=torch.float32, device=dev)
        self.momenta = torch.zeros(total, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, off in zip(self.params, offs):
                size, stride = tuple(p.shape), p.stride()
                new = torch.as_strided(self.values, size, stride, off)
                new.copy_(p.data)
                p.data = new
                p.grad = torch.as_strided(self.grads, size, stride, off)
                p._direct_grad = True

    def slice_bounds(self, first, last):
        """[start, end) floats covering params[first..last]."""
        end = self.offsets[last + 1] if last + 1 < len(self.params) else self.size
        return self.offsets[first], end


def _is_dense(p):
    # non-overlapping and dense: sorted strides multiply up to numel
    dims = sorted(zip(p.stride(), p.shape))
    expect = 1
    for st, sz in dims:
        if sz == 1:
            continue
        if st != expect:
            return False
        expect *= sz
    return True


class MomentumSGD(object):

    def __init__(self, lr=0.01, momentum=0.9):
        self.lr = lr
        self.momentum = momentum
        self.weight_decay = 0.
        self.target = None
        self.arena = None
        self.grad_sync = None      # set by parallel.DataParallelGradSync
        self.t = 0

    def setup(self, link):
        self.target = link
        return self

    def add_hook(self, hook):
        if isinstance(hook, WeightDecay):
            self.weight_decay = hook.rate
        else:
            raise TypeError('unsupported optimizer hook: %r' % (hook,))

    def _build(self):
        seen, params = set(), []
        for p in self.target.parameters():
            if p.requires_grad and id(p) not in seen:
                seen.add(id(p))
                params.append(p)
        if not params:
            raise ValueError('no trainable parameters')
        params.reverse()           # backward produces gradients in this order
        self.arena = ParamArena(params)
        if self.grad_sync is not None:
            self.grad_sync.attach(self)

    def update(self, lossfun=None, *args, **kwds):
        """One training iteration: loss = lossfun(*args); backward; (all-reduce); step."""
        if self.arena is None:
            self._build()
        loss = None
        if lossfun is not None:
            loss = lossfun(*args, **kwds)
            if self.grad_sync is not None:
                self.grad_sync.begin_backward()
            loss.backward()
        from .functions.conv import join_wgrad_stream
        join_wgrad_stream()            # weight gradients queued on the side stream
        scale = 1.0
        if self.grad_sync is not None:
            scale = self.grad_sync.finish()
        self.step(scale)
        return loss

    def step(self, grad_scale=1.0):
        a = self.arena
        _lib.call('mrcnn_sgd_momentum_wd', _lib.ptr(a.values), _lib.ptr(a.grads),
                  _lib.ptr(a.momenta), a.size, float(self.lr), float(self.momentum),
                  float(self.weight_decay), float(grad_scale), _lib.stream_ptr())
        self.t += 1
