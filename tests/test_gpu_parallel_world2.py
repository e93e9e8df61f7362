"""The data-parallel path at world size 2 ON THE DEVICE, with both ranks sharing the one GPU of the
test box (parallel.rehearsal(): MRCNN_DP_REHEARSAL=1 -> process group on gloo, which moves device
tensors through the host; RCCL refuses two ranks on one device).  Everything around the RCCL calls
runs as it does on an 8-GPU node — rendezvous from the environment, rank-0 broadcast of parameters
and buffers, gradient buckets polled from inside a real backward, the deferred weight gradients'
own reduction in the next step's proposal window, 1/world in the SGD launch — and `bench.py --gpus 2`
is driven end to end.  Replaces /root/reference/examples/train_common.py:96-104,178 (ChainerMN
communicator + multi-node optimizer); the arithmetic statement is chainermn's: after a step every
rank holds w0 - lr (mean over ranks of its gradient + weight decay)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(torch.cuda.is_available() and torch.cuda.device_count() >= 2,
                                 reason='a node with a GPU per rank runs tests/test_gpu_parallel_rccl.py instead: '
                                        'the rehearsal mode (two ranks on one device over gloo) refuses to start there')]

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LR, WD = 0.002, 1e-4


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world), MRCNN_DP_REHEARSAL='1')
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from chainer_mask_rcnn_amd import optimizers, parallel
    from chainer_mask_rcnn_amd.functions import conv
    from test_gpu_model import _build, freeze_like_reference
    r, w, local = parallel.init_from_env()
    assert (r, w, local) == (rank, world, 0) and dist.get_backend() == 'gloo'
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    conv.WINOGRAD_MIN_WORK = 1 << 24          # as tests/conftest.py: the full-size routes on the small model
    torch.manual_seed(10 + rank)              # different weights per rank until the broadcast
    model, chain, imgs, bboxes, labels, masks = _build(dev)
    with torch.no_grad():
        for p in model.parameters():
            p.add_(0.01 * rank)               # _build seeds its own weights: make the ranks differ
    opt = optimizers.MomentumSGD(lr=LR, momentum=0.9)
    opt.setup(chain)
    opt.add_hook(optimizers.WeightDecay(WD))
    freeze_like_reference(model, chain)
    a5 = model.head.res5.a
    opt.defer_weight_gradients([a5.conv2.W, a5.conv1.W])

    local_grads = {}

    class Recording(parallel.TorchDistExchange):
        def allreduce_async(self, tensor, bucket_id=0):
            torch.cuda.synchronize()          # (the slice is final once everything queued has run)
            local_grads[(tensor.data_ptr(), tensor.numel())] = tensor.detach().cpu().numpy().copy()
            super().allreduce_async(tensor, bucket_id)

    sync = parallel.DataParallelGradSync(opt, exchange=Recording(), bucket_bytes=4 << 20)
    opt._build()                              # arena + attach: rank-0 broadcast, bucket plan, hooks
    arena = opt.arena
    w0 = arena.values.detach().cpu().numpy().copy()
    base = arena.grads.data_ptr()
    # rank-dependent batch: rank 1 sees the mirrored images (boxes untouched: any batch will do)
    x = torch.tensor(imgs[:, :, :, ::-1].copy() if rank else imgs, device=dev)
    np.random.seed(5 + rank)
    loss = opt.update(chain, x, bboxes, labels, masks, [1., 1.])
    opt.flush()                               # the held-back gradients: reduced and applied
    torch.cuda.synchronize()
    w1 = arena.values.detach().cpu().numpy()
    # every slice that was exchanged, as this rank produced it -> gather -> expected update
    mine = np.zeros_like(w0)
    covered = np.zeros(w0.shape, bool)
    for (ptr, n), g in local_grads.items():
        off = (ptr - base) // 4
        mine[off:off + n] = g
        covered[off:off + n] = True
    gathered = [None] * world
    dist.all_gather_object(gathered, (mine, covered, float(loss.detach()), w0[:1000].copy(), w1.copy()))
    ok = {}
    ok['covered every trainable slice'] = bool(all(c.all() for _, c, _, _, _ in gathered))
    ok['same start after the broadcast'] = bool(all(np.array_equal(g[3], gathered[0][3]) for g in gathered))
    ok['ranks bit-identical after the step'] = bool(all(np.array_equal(g[4], gathered[0][4]) for g in gathered))
    ok['ranks saw different batches'] = gathered[0][2] != gathered[1][2]
    mean_g = sum(g[0].astype(np.float64) for g in gathered) / world
    step = -LR * (mean_g + WD * w0.astype(np.float64))
    # (the update is one fp32 fma chain per element: 1e-4 of the largest step + two ulps of the weights)
    err = np.abs((w1.astype(np.float64) - w0) - step).max() / (1e-4 * np.abs(step).max() + 2e-7 * np.abs(w0).max())
    ok['w1 == w0 - lr (mean gradient + wd w0)'] = bool(err <= 1.0 and np.abs(step).max() > 0)
    ok['finite'] = bool(np.isfinite(w1).all() and np.isfinite(float(loss.detach())))
    q.put((rank, ok, float(err), sync.describe()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_sharing_the_gpu_average_their_gradients(dev):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=10) for _ in range(2))
    for rank, ok, err, desc in res:
        assert all(ok.values()), (rank, ok, err)
        assert desc['ranks'] == 2 and 'gloo' in desc['library']
    print('world 2 on one GPU: update == w0 - lr (mean gradient + wd w0) (error / bound = %.2f)' % res[0][2])


@pytest.mark.parametrize('launcher', ['self-launch', 'torch.distributed.run'])
def test_bench_launch_path_at_two_ranks(dev, launcher):
    """`python bench.py --gpus 2` (bench starts its own ranks) and the driver's form
    (`python -m torch.distributed.run ... bench.py --gpus 2`), two ranks on this box's one GPU."""
    env = dict(os.environ, MRCNN_DP_REHEARSAL='1')
    tail = ['--gpus', '2', '--steps', '2', '--warmup', '1', '--repeats', '2', '--height', '320', '--width', '448']
    if launcher == 'self-launch':
        cmd = [sys.executable, os.path.join(ROOT, 'bench.py')] + tail
    else:
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
               '--master-addr', '127.0.0.1', '--master-port', str(_free_port()),
               os.path.join(ROOT, 'bench.py')] + tail
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-500:]                       # ONE line, from rank 0
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['config']['parallelism'] == 'dp2' and d['config']['global_batch'] == 4
    assert 'rehearsal' in d and d['config']['collective']['ranks'] == 2
    assert d['repeats']['n'] == 2 and np.isfinite(d['value']) and d['value'] > 0
    assert d['config']['loss'] is not None and np.isfinite(d['config']['loss'])
    assert 'rotating_h2d' not in d and 'fp32_mfma' not in d       # multi-GPU runs time the headline only
