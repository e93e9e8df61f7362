"""MomentumSGD + WeightDecay as the reference configures them
(/root/reference/examples/train_common.py:176-190; chainer rules):

    g += rate * p ;  v = momentum * v - lr * g ;  p += v      for every enabled parameter

All enabled parameters live in ONE flat fp32 arena (values, gradients and momenta), so
the update is a single HIP launch and the data-parallel gradient exchange is a handful
of large contiguous RCCL all-reduces instead of one message per parameter.  Weight
gradients are written by the wgrad kernels directly into the gradient arena.

Gradient ownership rules (chainer semantics, ``Optimizer.update`` / ``GradientMethod.update``):

* the gradient arena is cleared by the SGD launch itself (``cleargrads()`` of the next
  iteration), so nothing stale survives a step;
* the FIRST gradient a parameter receives in a step is written in place by the producing
  kernel (``ParamArena.claim``); any further gradient of the same parameter before the next
  step (a link used twice, two backward passes per update) goes through autograd's ordinary
  accumulation into the same arena view;
* a parameter that received NO gradient in a step is skipped entirely — no weight decay,
  no momentum decay — exactly like chainer skips ``param.grad is None``;
* ``p.grad`` must keep aliasing the arena: ``update()`` re-binds it if the caller replaced or
  dropped it (``zero_grad(set_to_none=True)``), folding a foreign gradient in first.
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
        self.epoch = 1            # bumped by every optimizer step
        with torch.no_grad():
            for i, (p, off) in enumerate(zip(self.params, offs)):
                size, stride = tuple(p.shape), p.stride()
                new = torch.as_strided(self.values, size, stride, off)
                new.copy_(p.data)
                p.data = new
                p.grad = self.grad_view(i)
                p._arena, p._arena_index, p._grad_epoch = self, i, 0
                p._direct_grad = True
                # gradients that reach the parameter through autograd's own accumulation
                p.register_post_accumulate_grad_hook(_mark_written)

    def grad_view(self, i):
        p = self.params[i]
        return torch.as_strided(self.grads, tuple(p.shape), p.stride(), self.offsets[i])

    def aliases(self, i):
        g = self.params[i].grad
        return g is not None and g.data_ptr() == self.grads.data_ptr() + 4 * self.offsets[i] \
            and g.stride() == self.params[i].stride()

    def claim(self, p):
        """True if the caller may WRITE (not add) p's gradient into ``p.grad`` now: the arena
        view is intact and nobody has produced a gradient for p yet in this step."""
        if p._grad_epoch == self.epoch or not self.aliases(p._arena_index):
            return False
        p._grad_epoch = self.epoch
        return True

    def rebind(self):
        """Restore ``p.grad`` = arena view for every parameter (folding in a gradient tensor
        that autograd allocated elsewhere because the view had been dropped)."""
        with torch.no_grad():
            for i, p in enumerate(self.params):
                if self.aliases(i):
                    continue
                view = self.grad_view(i)
                if p.grad is not None:
                    view.add_(p.grad)
                    p._grad_epoch = self.epoch
                p.grad = view

    def written(self):
        """Per parameter: did it receive a gradient since the last step?"""
        return [p._grad_epoch == self.epoch for p in self.params]

    def slice_bounds(self, first, last):
        """[start, end) floats covering params[first..last]."""
        end = self.offsets[last + 1] if last + 1 < len(self.params) else self.size
        return self.offsets[first], end


def _mark_written(p):
    p._grad_epoch = p._arena.epoch


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
        never, affine = _gradient_free_parameters(self.target)
        seen, params = set(), []
        for n, p in self.target.named_parameters():
            if not p.requires_grad or id(p) in seen:
                continue
            seen.add(id(p))
            if id(p) in affine:
                raise ValueError(
                    'AffineChannel2D parameter %r is enabled for update, but this build fuses '
                    'the affine into the convolution epilogue and produces no gradient for it '
                    '(the reference freezes every AffineChannel2D, examples/train_common.py:'
                    '188-190): call optimizers.disable_update() on it' % n)
            if id(p) in never:
                continue       # below freeze_at: grad is None in chainer -> never updated
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
        self.step(scale, zero_grads=True)
        return loss

    def step(self, grad_scale=1.0, zero_grads=False):
        """Apply the update rule to every parameter that received a gradient since the last
        step (one launch when that is all of them: the normal case).  ``zero_grads`` clears
        the gradient arena in the same pass."""
        a = self.arena
        a.rebind()
        written = a.written()
        if self.grad_sync is not None:
            # after the all-reduce every slice holds the sum over ranks (zeros from a rank whose
            # backward skipped the parameter): all ranks update everything, no host round trip
            written = [True] * len(written)
        runs = _runs(written)
        for first, last in runs:
            lo, hi = a.slice_bounds(first, last)
            _lib.call('mrcnn_sgd_momentum_wd_ex', _lib.ptr(a.values[lo:hi]),
                      _lib.ptr(a.grads[lo:hi]), _lib.ptr(a.momenta[lo:hi]), hi - lo,
                      float(self.lr), float(self.momentum), float(self.weight_decay),
                      float(grad_scale), 1 if zero_grads else 0, _lib.stream_ptr())
        if zero_grads and runs != [(0, len(a.params) - 1)]:
            a.grads.zero_()              # slices of skipped parameters (normally there are none)
        a.epoch += 1
        self.t += 1


def _runs(flags):
    """[(first, last)] index runs of consecutive True entries."""
    runs, start = [], None
    for i, f in enumerate(flags):
        if f and start is None:
            start = i
        elif not f and start is not None:
            runs.append((start, i - 1))
            start = None
    if start is not None:
        runs.append((start, len(flags) - 1))
    return runs


def _gradient_free_parameters(target):
    """(ids of parameters that can never receive a gradient here, ids of AffineChannel2D
    parameters).  The former are the extractor layers up to ``freeze_at``, which run under
    no_grad (``unchain_backward`` in models/resnet_extractor.py:86-87 of the reference)."""
    from .links import AffineChannel2D
    never, affine = set(), set()
    for m in target.modules():
        if isinstance(m, AffineChannel2D):
            affine.update(id(p) for p in m.parameters())
        freeze_at = getattr(m, 'freeze_at', None)
        if freeze_at is not None and hasattr(m, 'functions'):
            owners = {'conv1': ('conv1', 'bn1')}
            for key in m.functions:
                for attr in owners.get(key, (key,)):
                    sub = getattr(m, attr, None)
                    if isinstance(sub, torch.nn.Module):
                        never.update(id(p) for p in sub.parameters())
                if key == freeze_at:
                    break
    return never, affine
