"""N>1 path on CPU: world_size-2 gloo runs of the gradient-exchange logic used by the
data-parallel train step (bucketed all-reduce over the flat gradient arena, rank-0 weight
broadcast, 1/world scaling).  The HIP kernels are not involved (no GPU here)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ['RANK'] = str(rank)
    os.environ['WORLD_SIZE'] = str(world)
    os.environ['LOCAL_RANK'] = str(rank)
    from chainer_mask_rcnn_amd import parallel
    r, w, _ = parallel.init_from_env(backend='gloo')
    assert (r, w) == (rank, world)
    torch.manual_seed(rank)
    flat = torch.arange(1000, dtype=torch.float32) * (rank + 1)
    bounds = [(0, 300), (300, 304), (304, 1000)]
    buckets = parallel.GradBuckets(flat, bounds)
    buckets.launch(0)                 # issued early (during "backward")
    buckets.launch(0)                 # idempotent
    buckets.wait_all()                # remaining buckets + wait
    expect = torch.arange(1000, dtype=torch.float32) * sum(range(1, world + 1))
    ok = torch.equal(flat, expect)
    # second step re-uses the object
    flat.fill_(float(rank))
    buckets.launch(2)
    buckets.wait_all()
    ok = ok and torch.equal(flat, torch.full((1000,), float(sum(range(world)))))
    # rank-0 broadcast of the weights
    vals = torch.full((10,), float(rank + 7))
    dist.broadcast(vals, src=0)
    ok = ok and torch.equal(vals, torch.full((10,), 7.))
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_bucketed_allreduce_gloo_world2():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = dict(q.get(timeout=10) for _ in range(2))
    assert res == {0: True, 1: True}


def test_arena_layout_and_stage_buckets():
    """ParamArena: 16-byte aligned slices in reverse registration order; values/grads are
    views; DataParallelGradSync derives contiguous head+rpn / res4 / res3 buckets."""
    from chainer_mask_rcnn_amd import optimizers
    torch.manual_seed(0)

    class Tiny(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a = torch.nn.Parameter(torch.randn(3, 5))
            self.b = torch.nn.Parameter(torch.randn(7))
            w = torch.randn(4, 2, 2, 6).permute(0, 3, 1, 2)       # channels-last filter
            self.c = torch.nn.Parameter(w)
            self.frozen = torch.nn.Parameter(torch.randn(9), requires_grad=False)
    m = Tiny()
    ref = {n: p.detach().clone() for n, p in m.named_parameters()}
    params = [p for p in m.parameters() if p.requires_grad][::-1]
    arena = optimizers.ParamArena(params)
    assert arena.size % 4 == 0 and all(o % 4 == 0 for o in arena.offsets)
    assert arena.offsets == [0, 96, 104] and arena.size == 120
    for n, p in m.named_parameters():
        assert torch.equal(p, ref[n])
        if p.requires_grad:
            assert p.data_ptr() >= arena.values.data_ptr()
            assert p.grad is not None and p.grad.shape == p.shape and p._direct_grad
            assert p.stride() == ref[n].stride()
    m.c.grad.fill_(2.)
    assert float(arena.grads[:96].sum()) == 192.
    assert arena.slice_bounds(0, 1) == (0, 104)


def _sync_worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ['RANK'] = str(rank)
    os.environ['WORLD_SIZE'] = str(world)
    from chainer_mask_rcnn_amd import parallel, optimizers
    parallel.init_from_env(backend='gloo')

    class Blk(torch.nn.Module):
        def __init__(self, n):
            super().__init__()
            self.W = torch.nn.Parameter(torch.full((n,), float(rank + 1)))

    class Ext(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.res3, self.res4 = Blk(8), Blk(12)
            self.stage_hooks = {}

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.extractor, self.rpn, self.head = Ext(), Blk(5), Blk(20)

    class Chain(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.mask_rcnn = Net()
            self.features_grad_hook = None

    chain = Chain()
    opt = optimizers.MomentumSGD(lr=0.1)
    opt.setup(chain)
    sync = parallel.DataParallelGradSync(opt)
    opt._build()                                   # arena + attach (broadcast, buckets)
    a = opt.arena
    ok = sync.world_size == world
    # rank-0 weights everywhere
    ok = ok and bool((a.values[a.values != 0] == 1.0).all())
    # buckets: head+rpn | res4 | res3, contiguous, in backward order
    ok = ok and len(sync.buckets.bounds) == 3 and sync.buckets.bounds[0][0] == 0
    ok = ok and sync.buckets.bounds[-1][1] == a.size
    ok = ok and chain.features_grad_hook is not None and 'res3' in chain.mask_rcnn.extractor.stage_hooks
    # a "backward": every rank writes rank+1 into its gradients, hooks fire in stage order
    a.grads.fill_(float(rank + 1))
    chain.features_grad_hook(torch.zeros(1))
    chain.mask_rcnn.extractor.stage_hooks['res3'](torch.zeros(1))
    scale = sync.finish()
    ok = ok and abs(scale - 1.0 / world) < 1e-12
    ok = ok and bool((a.grads == float(sum(range(1, world + 1)))).all())
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_data_parallel_grad_sync_gloo_world2():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sync_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = dict(q.get(timeout=10) for _ in range(2))
    assert res == {0: True, 1: True}
