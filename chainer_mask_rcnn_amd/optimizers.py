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

Deferred weight gradients (opt-in, ``defer_weight_gradients``): the weight gradients of chosen
RoI-head layers and the update of exactly those parameters are held back at the end of a step
and run on a second stream inside the NEXT step's proposal window (between the RPN convolution
and the RoI head, where the GPU is otherwise nearly idle).  The parameters are updated before
they are read again, so every step computes the same values; ``flush()`` forces pending work
(called by ``predict``, the serializers and at the end of a timed region).
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
        self.deferred_params = []  # see defer_weight_gradients
        self._pending = None       # (jobs, [(lo, hi)], lr, momentum, wd, grad_scale)
        self._join = None          # event the compute stream must wait for before the head

    def setup(self, link):
        self.target = link
        return self

    def add_hook(self, hook):
        if isinstance(hook, WeightDecay):
            self.weight_decay = hook.rate
        else:
            raise TypeError('unsupported optimizer hook: %r' % (hook,))

    def defer_weight_gradients(self, params):
        """Hold the weight gradients (and the update) of ``params`` — parameters of the RoI
        head only: they must not be read between the end of a step and the next step's RoI
        head — back into the next step's proposal window.  Under data parallelism their
        gradient slices are all-reduced there too (call this BEFORE the first update: the
        gradient buckets are planned around them)."""
        self.deferred_params = list(params)
        if self not in _DEFERRING:
            _DEFERRING.append(self)

    # -- pending work ------------------------------------------------------------------------
    def launch_pending(self):
        """Queue the held-back weight gradients + their SGD slices on the defer stream, ordered
        after everything queued so far on the compute stream."""
        if self._pending is None:
            return
        from .functions import conv
        jobs, runs, lr, momentum, wd, scale = self._pending
        self._pending = None
        a = self.arena
        dev = a.values.device
        side, main = conv.defer_stream(dev), torch.cuda.current_stream(dev)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            conv.run_deferred_wgrads(jobs)
            if self.grad_sync is not None:          # data parallel: sum over ranks first
                self.grad_sync.reduce_deferred([a.grads[lo:hi] for lo, hi in runs])
            for lo, hi in runs:
                _lib.call('mrcnn_sgd_momentum_wd_ex', _lib.ptr(a.values[lo:hi]),
                          _lib.ptr(a.grads[lo:hi]), _lib.ptr(a.momenta[lo:hi]), hi - lo,
                          float(lr), float(momentum), float(wd), float(scale), 1, _lib.stream_ptr())
            conv.weights_changed()
            for job in jobs:                       # keep the operands alive until S2 is done
                for t in job[1:]:
                    if isinstance(t, torch.Tensor):
                        t.record_stream(side)
            ev = torch.cuda.Event()
            ev.record(side)
        self._join = ev

    def join_pending(self):
        """The compute stream waits for the deferred work (before the parameters are read)."""
        if self._join is not None:
            torch.cuda.current_stream(self.arena.values.device).wait_event(self._join)
            self._join = None

    def flush(self):
        self.launch_pending()
        self.join_pending()

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
        defer = None
        if lossfun is not None:
            loss = lossfun(*args, **kwds)      # (launches / joins the previous step's deferred work)
            self.flush()                       # nothing may stay pending across a backward
            if self.grad_sync is not None:
                self.grad_sync.begin_backward()
            from .functions import conv
            if self.deferred_params and (self.grad_sync is None or
                                         getattr(self.grad_sync, 'supports_deferred', False)):
                defer = conv._DEFER = conv.DeferQueue(self.deferred_params)
            try:
                loss.backward()
            finally:
                conv._DEFER = None
        from .functions.conv import join_wgrad_stream
        join_wgrad_stream()            # weight gradients queued on the side stream
        scale = 1.0
        if self.grad_sync is not None:
            scale = self.grad_sync.finish()
        self.step(scale, zero_grads=True, deferred=defer)
        return loss

    def step(self, grad_scale=1.0, zero_grads=False, deferred=None):
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
        held = set()
        if deferred is not None and self.grad_sync is not None:
            # Data parallel: a deferred parameter is in NO in-backward bucket (the bucket plan
            # leaves it to reduce_deferred).  If its weight gradient was not actually held back
            # this step (the in-place claim failed: gradient view dropped, parameter used twice,
            # a route that bypasses _wgrad_raw), nothing would all-reduce it and the ranks would
            # drift apart silently — fail loudly instead.
            queued = set(j[3].data_ptr() for j in deferred.jobs)
            missing = [i for i, p in enumerate(a.params)
                       if id(p) in deferred.ids and p._grad_epoch == a.epoch
                       and p.grad.data_ptr() not in queued]
            if missing:
                raise RuntimeError(
                    'defer_weight_gradients: %d deferred parameter(s) received a gradient that was '
                    'not held back (arena index %s); under data parallelism their gradients would '
                    'never be all-reduced.  Defer only RoI-head convolution filters that are used '
                    'once per step.' % (len(missing), missing[:4]))
        if deferred is not None and deferred.jobs:
            # parameters whose gradient kernels were held back: their slices are updated by
            # launch_pending(), after those kernels, in the next step's proposal window
            grads = set(j[3].data_ptr() for j in deferred.jobs)
            held = set(i for i, p in enumerate(a.params) if p.grad.data_ptr() in grads)
            self._pending = (deferred.jobs,
                             [a.slice_bounds(f, l) for f, l in _runs([i in held for i in range(len(a.params))])],
                             self.lr, self.momentum, self.weight_decay, grad_scale)
        all_written = all(written)
        runs = _runs([w and i not in held for i, w in enumerate(written)])
        for first, last in runs:
            lo, hi = a.slice_bounds(first, last)
            _lib.call('mrcnn_sgd_momentum_wd_ex', _lib.ptr(a.values[lo:hi]),
                      _lib.ptr(a.grads[lo:hi]), _lib.ptr(a.momenta[lo:hi]), hi - lo,
                      float(self.lr), float(self.momentum), float(self.weight_decay),
                      float(grad_scale), 1 if zero_grads else 0, _lib.stream_ptr())
        from .functions import conv
        conv.weights_changed()       # the kernel writes the arena behind torch's version counters
        if zero_grads and not all_written:
            a.grads.zero_()              # slices of skipped parameters (normally there are none)
        a.epoch += 1
        self.t += 1


_DEFERRING = []       # optimizers that may hold deferred work (see defer_weight_gradients)
# where the held-back work is queued: 'window-open' = right behind the RPN convolutions, beside
# the top-k / NMS kernels (measured best: 49.6-50.0 vs 50.6-51.2 ms per step without deferral);
# 'after-proposals' = once the proposal read-back has returned (50.0-50.6).  Running the proposal
# kernels on a separate high-priority stream next to the deferred work was measured and dropped
# (64 ms per step).
import os as _os
DEFER_LAUNCH_AT = _os.environ.get('MRCNN_DEFER_AT', 'window-open')


def launch_pending_all():
    """Called by the RPN right after its convolutions are queued: the proposal window opens."""
    for opt in _DEFERRING:
        opt.launch_pending()


def join_pending_all():
    """Called by the RoI head before it reads its parameters."""
    for opt in _DEFERRING:
        opt.join_pending()


def flush_all():
    """Force every pending deferred update (anything that reads parameters outside a train
    step: predict, snapshots)."""
    for opt in _DEFERRING:
        if opt.arena is not None:
            opt.flush()


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
