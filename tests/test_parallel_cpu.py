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
    views of the flat arenas; first-gradient claim / epoch bookkeeping."""
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
    # first gradient of a step may be written in place, a second one must be accumulated
    assert arena.claim(m.a) and not arena.claim(m.a)
    assert arena.written() == [False, False, True]        # arena order: c, b, a
    arena.epoch += 1
    assert arena.written() == [False, False, False] and arena.claim(m.a)
    # a dropped / replaced .grad is detected and re-bound (its content folded in)
    m.b.grad = None
    m.a.grad = torch.ones_like(m.a)
    assert not arena.claim(m.b)
    arena.rebind()
    assert arena.aliases(0) and arena.aliases(1) and arena.aliases(2)
    assert float(m.a.grad.sum()) == 15. and m.a.grad.data_ptr() == arena.grads.data_ptr() + 4 * 104


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
            self.b = torch.nn.Parameter(torch.full((4,), float(rank + 1)))

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.res3, self.res4, self.rpn, self.head = Blk(8), Blk(12), Blk(5), Blk(20)
            self.frozen = torch.nn.Parameter(torch.full((6,), float(10 + rank)), requires_grad=False)
            self.register_buffer('stat', torch.full((3,), float(20 + rank)))

    class Chain(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.mask_rcnn = Net()
            self.features_grad_hook = None

    class Recording(parallel.TorchDistExchange):
        def __init__(self):
            super().__init__()
            self.log = []

        def allreduce_async(self, tensor, bucket_id=0):
            self.log.append(bucket_id)
            super().allreduce_async(tensor, bucket_id)

    chain = Chain()
    opt = optimizers.MomentumSGD(lr=0.1)
    opt.setup(chain)
    ex = Recording()
    sync = parallel.DataParallelGradSync(opt, exchange=ex, bucket_bytes=64)
    opt._build()                                   # arena + attach (broadcast, buckets)
    a = opt.arena
    net = chain.mask_rcnn
    ok = sync.world_size == world
    # bcast_data: rank-0 values everywhere, INCLUDING frozen parameters and buffers
    ok = ok and bool((a.values[a.values != 0] == 1.0).all())
    ok = ok and bool((net.frozen == 10.).all()) and bool((net.stat == 20.).all())
    # buckets: cut at module boundaries (W and b of one module never separated), contiguous,
    # in backward order (reverse registration: head, rpn, res4, res3), each >= 64 bytes
    bp = sync.bucket_params
    names = {id(p): n for n, p in chain.named_parameters()}
    owners = [[names[id(p)].rsplit('.', 1)[0] for p in a.params[lo:hi + 1]] for lo, hi in bp]
    ok = ok and owners == [['mask_rcnn.head', 'mask_rcnn.head'],
                           ['mask_rcnn.rpn', 'mask_rcnn.rpn', 'mask_rcnn.res4', 'mask_rcnn.res4'],
                           ['mask_rcnn.res3', 'mask_rcnn.res3']]
    ok = ok and sync.buckets.bounds[0][0] == 0 and sync.buckets.bounds[-1][1] == a.size
    ok = ok and all(sync.buckets.bounds[i][1] == sync.buckets.bounds[i + 1][0] for i in range(len(bp) - 1))
    ok = ok and chain.features_grad_hook is not None
    # a "backward": gradients appear head -> rpn -> res4 -> res3; a bucket is queued only once
    # EVERY parameter in it has a gradient, and always in order
    def write(mod):
        for p in mod.parameters():
            assert a.claim(p)
            p.grad.fill_(float(rank + 1))
    write(net.head); sync.poll()
    ok = ok and ex.log == [0]
    write(net.rpn); chain.features_grad_hook(torch.zeros(1))
    ok = ok and ex.log == [0]                      # res4 shares the bucket: not ready yet
    write(net.res4); sync.poll()
    ok = ok and ex.log == [0, 1]
    write(net.res3)                                # never polled: finish() launches the rest
    scale = sync.finish()
    ok = ok and ex.log == [0, 1, 2]
    ok = ok and abs(scale - 1.0 / world) < 1e-12
    total = float(sum(range(1, world + 1)))
    ok = ok and all(bool((p.grad == total).all()) for p in a.params)

    # held-back (deferred) parameters: left out of the in-backward buckets, reduced on their own
    chain2 = Chain()
    opt2 = optimizers.MomentumSGD(lr=0.1)
    opt2.setup(chain2)
    net2 = chain2.mask_rcnn
    opt2.defer_weight_gradients([net2.rpn.W, net2.rpn.b])
    ex2 = Recording()
    sync2 = parallel.DataParallelGradSync(opt2, exchange=ex2, bucket_bytes=64)
    opt2._build()
    a2 = opt2.arena
    held = set(id(p) for p in opt2.deferred_params)
    covered = set()
    for lo, hi in sync2.bucket_params:
        covered.update(range(lo, hi + 1))
    ok = ok and all((i in covered) != (id(p) in held) for i, p in enumerate(a2.params))
    a2.grads.fill_(float(rank + 1))
    for p in a2.params:
        if id(p) not in held:
            a2.claim(p)
    sync2.finish()                                  # every regular bucket, none of the held slices
    idx = [i for i, p in enumerate(a2.params) if id(p) in held]
    lo, hi = a2.slice_bounds(idx[0], idx[-1])
    ok = ok and bool((a2.grads[lo:hi] == float(rank + 1)).all())
    ok = ok and bool((a2.grads[:lo] == total).all()) and bool((a2.grads[hi:] == total).all())
    sync2.reduce_deferred([a2.grads[lo:hi]])        # what MomentumSGD.launch_pending does
    ok = ok and bool((a2.grads == total).all()) and ex2.log[-1] >= 1000
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


# ---- rehearsal of the 8-rank launch on the REAL ResNet50-C4 arena --------------------------------
# /root/reference/examples/train_common.py:96-104,160-190: one process per GPU, rank-0 parameters
# broadcast, gradients averaged before the update.  The driver's 8-GPU run is the first time the
# data-parallel path runs at world size 8; this is the same code — bench.build_trainer (model,
# frozen set, deferred res5 gradients, 16 MB buckets), DataParallelGradSync.attach, the in-backward
# polling, finish(), reduce_deferred — at world size 8 over gloo, with the exchange the only stand-in
# (TorchDistExchange instead of RCCL behind the C ABI; same interface).
def _rehearsal_worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ['RANK'] = str(rank)
    os.environ['WORLD_SIZE'] = str(world)
    os.environ['LOCAL_RANK'] = str(rank)
    torch.set_num_threads(1)
    import zlib
    import bench
    from chainer_mask_rcnn_amd import parallel
    r, w, _ = parallel.init_from_env(backend='gloo')
    failed = []

    def ck(k, cond):
        if not cond:
            failed.append(k)
    ck(0, (r, w) == (rank, world))
    torch.manual_seed(100 + rank)                      # every rank starts from DIFFERENT weights
    cpu = torch.device('cpu')
    model, chain, opt, sync = bench.build_trainer(50, cpu, world, 2 * world, bucket_bytes=16 << 20, defer=5)
    ck(1, sync is not None and isinstance(sync.exchange, parallel.TorchDistExchange))
    opt._build()                                       # arena + attach: broadcast, bucket plan, hooks
    a = opt.arena

    def checksum():
        h = 0
        for _, t in sorted(list(chain.named_parameters()) + list(chain.named_buffers()), key=lambda kv: kv[0]):
            h = zlib.crc32(t.detach().contiguous().numpy().tobytes(), h)
        return h
    sums = [None] * world
    dist.all_gather_object(sums, checksum())
    ck(2, len(set(sums)) == 1)                    # rank 0's parameters AND buffers everywhere
    # the plan: identical on every rank, 143 MB of gradients in ~16 MB buckets cut at block
    # boundaries, the five held-back res5 filters in none of them, everything else in exactly one
    plans = [None] * world
    dist.all_gather_object(plans, (sync.bucket_params, sync.buckets.bounds, a.size))
    ck(3, all(p == plans[0] for p in plans))
    held = set(id(p) for p in opt.deferred_params)
    ck(4, len(held) == 5)
    covered = []
    for lo, hi in sync.bucket_params:
        covered += list(range(lo, hi + 1))
    ck(5, len(covered) == len(set(covered)))
    ck(6, all((i in set(covered)) != (id(p) in held) for i, p in enumerate(a.params)))
    mb = [(e - s) * 4 / 2 ** 20 for s, e in sync.buckets.bounds]
    ck(7, 130 < a.size * 4 / 2 ** 20 < 160 and 4 <= len(mb) <= 12)
    # (only the bucket that ends a run of consecutive non-held parameters may be smaller than 16 MB)
    from chainer_mask_rcnn_amd.optimizers import _runs
    n_runs = len(_runs([id(p) not in held for p in a.params]))
    ck(8, sum(1 for m in mb if m < 16.0) <= n_runs and max(mb) < 64.0)
    from chainer_mask_rcnn_amd.models.resnet_extractor import BuildingBlock
    blocks = [m for m in chain.modules() if isinstance(m, BuildingBlock)]
    ck(9, len(blocks) == 4 and all(m.grad_poll == sync.poll for m in blocks))
    ck(10, chain.features_grad_hook is not None)
    # a backward in arena order (= the order backward produces gradients), polled as the fused stage
    # nodes do (after every few parameters); rank r contributes r + 1 everywhere
    log = []
    real = sync.exchange.allreduce_async
    sync.exchange.allreduce_async = lambda t, b=0: (log.append(b), real(t, b))[1]
    first_launch_at = None
    for i, p in enumerate(a.params):
        assert a.claim(p)
        p.grad.fill_(float(rank + 1))
        if i % 3 == 2:
            sync.poll()
            if first_launch_at is None and log:
                first_launch_at = i
    ck(11, first_launch_at is not None and first_launch_at < len(a.params) // 2)   # overlaps backward
    scale = sync.finish()
    ck(12, log == list(range(len(mb))) and abs(scale * world - 1.0) < 1e-12)
    total = float(sum(range(1, world + 1)))
    for i, p in enumerate(a.params):
        want = float(rank + 1) if id(p) in held else total
        ck(13, bool((p.grad == want).all()))
    # the held-back slices: reduced on their own, as MomentumSGD.launch_pending does
    runs = [a.slice_bounds(f, l) for f, l in _runs([id(p) in held for p in a.params])]
    sync.reduce_deferred([a.grads[lo:hi] for lo, hi in runs])
    ck(14, all(bool((p.grad == total).all()) for p in a.params))
    q.put((rank, failed, len(mb)))
    dist.barrier()
    dist.destroy_process_group()


def test_eight_rank_rehearsal_on_the_resnet50_arena():
    world = 8
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rehearsal_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    res = [q.get(timeout=10) for _ in range(world)]
    assert sorted(r[0] for r in res) == list(range(world))
    assert all(r[1] == [] for r in res), res      # (numbers of the failed checks per rank)
    assert len(set(r[2] for r in res)) == 1


def test_rehearsal_mode_refuses_a_node_with_a_gpu_per_rank(monkeypatch, capsys):
    """A leaked MRCNN_DP_REHEARSAL=1 must not silently put every rank on device 0 of a real multi-GPU
    node (parallel.init_from_env): refused when >= WORLD_SIZE devices are visible, announced on stderr
    otherwise."""
    from chainer_mask_rcnn_amd import parallel
    monkeypatch.setenv('MRCNN_DP_REHEARSAL', '1')
    monkeypatch.setenv('WORLD_SIZE', '2')
    monkeypatch.setenv('RANK', '0')
    monkeypatch.setattr(torch.cuda, 'is_available', lambda: True)
    monkeypatch.setattr(torch.cuda, 'device_count', lambda: 8)
    with pytest.raises(RuntimeError, match='FEWER GPUs than'):
        parallel.init_from_env()
    # fewer devices than ranks: allowed, and loud
    monkeypatch.setattr(torch.cuda, 'device_count', lambda: 1)
    monkeypatch.setattr(parallel.dist, 'is_initialized', lambda: True)      # (no rendezvous in this test)
    rank, world, local = parallel.init_from_env()
    assert (rank, world, local) == (0, 2, 0)
    assert 'MRCNN_DP_REHEARSAL=1' in capsys.readouterr().err
