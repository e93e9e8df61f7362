"""Developer tool: GPU-side phase timeline of one train step (HIP events on the compute stream at
the phase boundaries): extractor fwd | RPN convs | proposal window | RoI head fwd | losses |
head backward (+RPN backward) | backbone backward + SGD.  Events are recorded where the host
QUEUES the boundary, so a phase is the GPU time between two boundaries of the compute stream."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from chainer_mask_rcnn_amd import _lib


def main():
    dev = torch.device('cuda:0')
    import random
    random.seed(0); np.random.seed(0); torch.manual_seed(0)
    rng = np.random.RandomState(0)
    layers = int(os.environ.get('LAYERS', 50))
    imgs, bboxes, labels, masks, scales = bench.synthetic_batch(rng, 2, 800, 1333)
    model, chain, opt, sync = bench.build_trainer(layers, dev, 1, 2, defer=int(os.environ.get('DEFER', 5)))
    for kv in os.environ.get('BENCH_TUNE', '').split(','):
        if '=' in kv:
            k, v = kv.split('=')
            _lib.check(_lib.load().mrcnn_set_tuning(k.encode(), int(v)), 'set_tuning')
    imgs_d = torch.tensor(imgs, device=dev).contiguous(memory_format=torch.channels_last)
    if int(os.environ.get('PREFETCH', 1)):
        chain.next_imgs = imgs_d
    marks = []

    def mark(label):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        marks.append((label, ev))

    ext_fwd, rpn_fwd, head_fwd = model.extractor.forward, model.rpn.forward, model.head.forward
    conv1_fwd = model.rpn.loc_score.forward

    def ext(x):
        mark('step begin')
        f = ext_fwd(x)
        mark('extractor fwd')
        if f.requires_grad:
            f.register_hook(lambda g: (mark('head+rpn bwd'), g)[1])
        return f

    def loc_score(h, *a, **k):
        out = conv1_fwd(h, *a, **k)
        mark('rpn convs fwd')
        return out

    def rpn(*a, **k):
        out = rpn_fwd(*a, **k)
        mark('proposals')
        return out

    def head(*a, **k):
        mark('roi sampling (host)')
        out = head_fwd(*a, **k)
        mark('head fwd')
        return out

    model.extractor.forward, model.rpn.forward, model.head.forward = ext, rpn, head
    model.rpn.loc_score.forward = loc_score
    chain_fwd = chain.forward

    def chain_f(*a, **k):
        loss = chain_fwd(*a, **k)
        mark('losses')
        return loss
    chain.forward = chain_f

    for _ in range(4):
        opt.update(chain, imgs_d, bboxes, labels, masks, scales)
    torch.cuda.synchronize()
    n = 6
    acc, order = {}, []
    import time
    t0 = time.perf_counter()
    for _ in range(n):
        del marks[:]
        opt.update(chain, imgs_d, bboxes, labels, masks, scales)
        mark('backbone bwd + sgd')
        torch.cuda.synchronize()
        for (la, ea), (lb, eb) in zip(marks[:-1], marks[1:]):
            if lb not in acc:
                order.append(lb)
            acc[lb] = acc.get(lb, 0.) + ea.elapsed_time(eb)
    wall = (time.perf_counter() - t0) / n * 1e3
    tot = 0.
    for k in order:
        print('%-24s %7.2f ms' % (k, acc[k] / n))
        tot += acc[k] / n
    print('%-24s %7.2f ms   (wall incl. per-step drain %.2f)' % ('sum', tot, wall))


if __name__ == '__main__':
    main()
