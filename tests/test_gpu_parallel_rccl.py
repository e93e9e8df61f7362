"""The first MULTI-RANK run of the RCCL path as a test (not as a benchmark): N = min(visible GPUs, 8)
processes, one per GPU, through ``RcclExchange`` — i.e. ``csrc/comm.hip`` behind ``mrcnn_allreduce_*`` —
with NO rehearsal flag.  Skipped on a 1-GPU box (the skip reason is printed in the test report); runs the
day ``pytest -m gpu`` lands on a multi-GPU node.  Replaces
/root/reference/examples/train_common.py:96-104,178 (ChainerMN communicator + multi-node optimizer)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

N_GPUS = torch.cuda.device_count() if torch.cuda.is_available() else 0
pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(N_GPUS < 2, reason='multi-rank RCCL needs >= 2 GPUs on this node (found %d): '
                                                    'RCCL refuses two ranks on one device' % N_GPUS)]

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LR, WD = 0.002, 1e-4


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    try:
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank),
                          LOCAL_RANK=str(rank), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY='0')
        os.environ.pop('MRCNN_DP_REHEARSAL', None)
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, 'tests'))
        from chainer_mask_rcnn_amd import optimizers, parallel
        from chainer_mask_rcnn_amd.functions import conv
        from test_gpu_model import _build, freeze_like_reference
        r, w, local = parallel.init_from_env()
        assert (r, w, local) == (rank, world, rank)
        torch.cuda.set_device(local)
        dev = torch.device('cuda', local)
        out = {}

        # (1) the C-ABI communicator: sum against the analytic value, broadcast of rank 0's buffer
        ex = parallel.RcclExchange(dev)
        assert ex.world_size == world and ex.describe()['ranks'] == world
        t = torch.full((1 << 20,), float(rank + 1), dtype=torch.float32, device=dev)
        ex.allreduce_async(t, 0)
        ex.wait_all()
        torch.cuda.synchronize()
        out['sum_ok'] = bool((t == world * (world + 1) / 2.0).all())
        b = torch.full((4099,), float(rank), dtype=torch.float32, device=dev)
        ex.broadcast(b, 0)
        torch.cuda.synchronize()
        out['bcast_ok'] = bool((b == 0).all())
        ex.barrier()
        ex.close()
        assert ex.handle is None                      # clean destroy; a second communicator follows

        # (2) two steps of the small model: different initial weights and batches per rank, the
        # rank-0 broadcast at attach(), buckets from inside backward, one deferred slice
        conv.WINOGRAD_MIN_WORK = 1 << 24
        model, chain, imgs, bboxes, labels, masks = _build(dev)
        with torch.no_grad():
            for p in model.parameters():
                p.add_(0.01 * rank)
        opt = optimizers.MomentumSGD(lr=LR, momentum=0.9)
        opt.setup(chain)
        opt.add_hook(optimizers.WeightDecay(WD))
        freeze_like_reference(model, chain)
        a5 = model.head.res5.a
        opt.defer_weight_gradients([a5.conv2.W, a5.conv3.W])
        sync = parallel.DataParallelGradSync(opt, bucket_bytes=4 << 20)
        assert isinstance(sync.exchange, parallel.RcclExchange), sync.exchange
        x = torch.tensor(imgs if rank % 2 == 0 else imgs[:, :, :, ::-1].copy(), device=dev)
        np.random.seed(5 + rank)
        losses = [float(opt.update(chain, x, bboxes, labels, masks, [1., 1.]).detach()) for _ in range(2)]
        opt.flush()
        torch.cuda.synchronize()
        w_ = opt.arena.values.detach().cpu().numpy()
        out['finite'] = bool(np.isfinite(losses).all() and np.isfinite(w_).all())
        # every rank ends with bit-identical weights: compare through the control plane
        mine = torch.tensor(np.frombuffer(w_.tobytes(), dtype=np.uint8).astype(np.int64).sum()
                            + int(np.abs(w_.view(np.int32).astype(np.int64)).sum() % (1 << 40)))
        lo, hi = mine.clone(), mine.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        out['identical'] = bool(lo.item() == hi.item())
        out['buckets'] = len(sync.buckets.bounds) if sync.buckets is not None else 0
        sync.exchange.close()
        q.put((rank, out, None))
    except Exception as e:          # noqa: BLE001  (reported to the parent, which fails the test)
        import traceback
        q.put((rank, None, traceback.format_exc()))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_multi_rank_rccl_allreduce_broadcast_and_two_train_steps():
    world = min(N_GPUS, 8)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
    for rank, out, err in sorted(results, key=lambda t: t[0]):
        assert err is None, 'rank %d:\n%s' % (rank, err)
        assert out['sum_ok'] and out['bcast_ok'], (rank, out)
        assert out['finite'] and out['identical'] and out['buckets'] >= 1, (rank, out)
    assert all(p.exitcode == 0 for p in procs)
