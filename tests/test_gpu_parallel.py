"""The data-parallel gradient path on the device: a 1-rank RCCL communicator created through
the C ABI (mrcnn_allreduce_*), buckets queued from inside backward on the library's collective
stream, SGD after the wait.  With one rank the all-reduce is the identity and 1/world = 1, so
losses and weights must be BIT-identical to the plain (non-DP) path after several steps — which
checks the stream/event ordering, the bucket cover of the arena and the poll points; the
multi-rank arithmetic is covered by the world-size-2 gloo tests (tests/test_parallel_cpu.py).
Replaces /root/reference/examples/train_common.py:96-104,178 (ChainerMN communicator +
multi-node optimizer)."""
import ctypes
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist

from chainer_mask_rcnn_amd import optimizers, parallel, _lib
from test_gpu_model import _build, freeze_like_reference

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def control_plane():
    """torch.distributed as the control plane only (the store carries the RCCL unique id)."""
    created = False
    if not dist.is_initialized():
        s = socket.socket()
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
        s.close()
        os.environ['MASTER_ADDR'] = '127.0.0.1'
        os.environ['MASTER_PORT'] = str(port)
        torch.cuda.set_device(0)
        dist.init_process_group('cpu:gloo,cuda:nccl', rank=0, world_size=1)   # as parallel.init_from_env
        created = True
    yield
    if created:
        dist.destroy_process_group()


def _run(dev, dp, steps=3, bucket_bytes=4 << 20, exchange=None, defer=False):
    model, chain, imgs, bboxes, labels, masks = _build(dev)
    opt = optimizers.MomentumSGD(lr=0.002, momentum=0.9)
    opt.setup(chain)
    opt.add_hook(optimizers.WeightDecay(1e-4))
    freeze_like_reference(model, chain)
    if defer:
        a = model.head.res5.a
        opt.defer_weight_gradients([a.conv2.W, a.conv1.W, a.conv3.W, a.conv4.W])
    sync = parallel.DataParallelGradSync(opt, bucket_bytes=bucket_bytes, exchange=exchange) if dp else None
    x = torch.tensor(imgs, device=dev)
    np.random.seed(5)
    losses = []
    for _ in range(steps):
        losses.append(opt.update(chain, x, bboxes, labels, masks, [1., 1.]).item())
    opt.flush()
    torch.cuda.synchronize()
    return losses, opt.arena.values.clone(), sync, opt


def test_rccl_abi_one_rank_roundtrip(dev, control_plane):
    """mrcnn_allreduce_unique_id / init / bucket / wait / broadcast / info / timing / destroy."""
    ex = parallel.RcclExchange()
    assert (ex.rank, ex.world_size) == (0, 1) and ex.rccl_version > 0
    d = ex.describe()
    assert d['ranks'] == 1 and 'RCCL' in d['library']
    t = torch.arange(1 << 16, dtype=torch.float32, device=dev)
    ref = t.clone()
    ex.timing(True)
    ex.allreduce_async(t[:1000], 0)
    ex.allreduce_async(t[1000:], 1)
    ex.wait_all()
    ex.broadcast(t, src=0)
    ex.barrier()
    assert torch.equal(t, ref)                      # one rank: sum == identity
    times = ex.bucket_times(2)
    assert [n for _, _, n in times] == [1, 1]
    assert times[0][1] == 4000. and times[1][1] == 4. * ((1 << 16) - 1000)
    ex.timing(False)
    # argument validation through the ABI (no abort, message available)
    lib = _lib.load()
    assert lib.mrcnn_allreduce_bucket(ex.handle, None, 16, 0, None, None) != 0
    assert b'null buffer' in lib.mrcnn_last_error()
    assert lib.mrcnn_allreduce_init(ctypes.create_string_buffer(128), 3, 2,
                                    ctypes.byref(ctypes.c_void_p())) != 0
    ex.close()


def test_data_parallel_one_rank_bit_identical(dev, control_plane):
    l_ref, w_ref, _, _ = _run(dev, dp=False)
    l_dp, w_dp, sync, opt = _run(dev, dp=True)
    assert all(np.isfinite(l_ref))
    assert l_dp == l_ref
    assert torch.equal(w_dp, w_ref)
    # the buckets cover the arena exactly once, in order, cut at layer-block boundaries
    b = sync.buckets.bounds
    assert b[0][0] == 0 and b[-1][1] == opt.arena.size and len(b) >= 4
    assert all(b[i][1] == b[i + 1][0] for i in range(len(b) - 1))
    names = {id(p): n for n, p in opt.target.named_parameters()}
    for lo, hi in sync.bucket_params:
        first, last = names[id(opt.arena.params[lo])], names[id(opt.arena.params[hi])]
        assert first and last
    # buckets were launched from inside backward (before finish()): the first one closes when
    # res5.b2's weight gradients are queued
    sync.exchange.close()


def test_buckets_launch_during_backward(dev, control_plane):
    """Poll points: at least the head / res5 buckets are queued while backward is still running
    (i.e. before MomentumSGD.update() reaches finish())."""
    model, chain, imgs, bboxes, labels, masks = _build(dev)
    opt = optimizers.MomentumSGD(lr=0.002, momentum=0.9)
    opt.setup(chain)
    freeze_like_reference(model, chain)
    sync = parallel.DataParallelGradSync(opt, bucket_bytes=4 << 20)
    x = torch.tensor(imgs, device=dev)
    np.random.seed(5)
    opt.update(chain, x, bboxes, labels, masks, [1., 1.])     # builds the arena, attaches
    seen = {}
    orig_finish = sync.finish

    def spying_finish():
        seen['launched_before_finish'] = sum(sync.buckets.launched)
        return orig_finish()
    sync.finish = spying_finish
    opt.update(chain, x, bboxes, labels, masks, [1., 1.])
    torch.cuda.synchronize()
    n = len(sync.buckets.bounds)
    assert seen['launched_before_finish'] >= n - 1, (seen, n)
    sync.exchange.close()


def test_fallback_exchange_one_rank_bit_identical(dev, control_plane):
    """The library fallback of parallel.default_exchange (torch.distributed's own RCCL binding,
    used only if mrcnn_allreduce_init fails on some rank) drives the same buckets and poll
    points: bit-identical to the plain path with one rank."""
    l_ref, w_ref, _, _ = _run(dev, dp=False)
    l_fb, w_fb, sync, _ = _run(dev, dp=True, exchange=parallel.TorchDistExchange())
    assert l_fb == l_ref and torch.equal(w_fb, w_ref)
    assert sync.describe()['library'].startswith('torch.distributed')


def test_deferred_weight_gradients_under_data_parallel(dev, control_plane):
    """Held-back weight gradients are all-reduced (C-ABI RCCL, on the defer stream) and applied
    in the next step's proposal window; the in-backward buckets are planned around them."""
    l_ref, w_ref, _, _ = _run(dev, dp=False)
    l_dp, w_dp, sync, opt = _run(dev, dp=True, defer=True)
    assert l_dp == l_ref and torch.equal(w_dp, w_ref)
    held = set(id(p) for p in opt.deferred_params)
    covered = set()
    for lo, hi in sync.bucket_params:
        covered.update(range(lo, hi + 1))
    for i, p in enumerate(opt.arena.params):
        assert (i in covered) != (id(p) in held)        # every parameter exactly one way
    sync.exchange.close()
