"""Developer tool: host-side timeline of one train step (where the host waits / works while
the GPU runs).  Marks are perf_counter stamps taken by MaskRCNNTrainChain.forward."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else 'host'   # host | device | device-masks
    dev = torch.device('cuda:0')
    import random
    random.seed(0); np.random.seed(0); torch.manual_seed(0)
    rng = np.random.RandomState(0)
    imgs, bboxes, labels, masks, scales = bench.synthetic_batch(rng, 2, 800, 1333)
    model, chain, opt, sync = bench.build_trainer(50, dev, 1, 2)
    imgs_d = torch.tensor(imgs, device=dev).contiguous(memory_format=torch.channels_last)
    chain.device_targets = mode != 'host'
    if mode == 'device-masks':       # ground-truth masks resident on the device (uint8)
        masks = [torch.tensor(np.asarray(m) != 0, device=dev).to(torch.uint8) for m in masks]
    print('target creators:', mode)
    for _ in range(3):
        opt.update(chain, imgs_d, bboxes, labels, masks, scales)
    torch.cuda.synchronize()
    acc = {}
    n = 5
    for _ in range(n):
        chain.host_timeline = []
        t0 = time.perf_counter()
        opt.update(chain, imgs_d, bboxes, labels, masks, scales)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        tl = chain.host_timeline + [('backward+sgd queued', t1), ('gpu drained', t2)]
        prev = t0
        for label, t in tl:
            acc[label] = acc.get(label, 0.) + (t - prev)
            prev = t
    for k, v in acc.items():
        print('%-24s %7.2f ms' % (k, v / n * 1e3))
    print('%-24s %7.2f ms' % ('total', sum(acc.values()) / n * 1e3))
    # wall-clock throughput of the same loop without the per-step drain
    chain.host_timeline = None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        opt.update(chain, imgs_d, bboxes, labels, masks, scales)
    torch.cuda.synchronize()
    print('%-24s %7.2f ms/step' % ('10 steps back to back', (time.perf_counter() - t0) * 100))


if __name__ == '__main__':
    main()
