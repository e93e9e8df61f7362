"""Developer check: does torch.distributed's gloo backend all-reduce / broadcast device tensors when two
ranks share ONE GPU (what a world-size-2 rehearsal of bench.py on a 1-GPU box needs)?"""
import os, sys
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def worker(rank, world, port):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), GLOO_SOCKET_IFNAME='lo')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.cuda.set_device(0)
    t = torch.full((1 << 20,), float(rank + 1), device='cuda:0')
    w = dist.all_reduce(t, async_op=True)
    w.wait()
    torch.cuda.synchronize()
    b = torch.full((8,), float(rank), device='cuda:0')
    dist.broadcast(b, src=0)
    print('rank', rank, 'allreduce ->', float(t[0]), float(t[-1]), 'broadcast ->', float(b[0]), flush=True)
    dist.destroy_process_group()


if __name__ == '__main__':
    mp.spawn(worker, args=(2, 29533), nprocs=2)
