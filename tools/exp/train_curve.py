"""Developer check: the train step under the default arithmetic really trains — loss over a few
hundred steps on the rotating synthetic batches — and is bit-reproducible run to run."""
import os, sys, random
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench


def run(steps, n_batches=4):
    dev = torch.device('cuda:0')
    random.seed(0); np.random.seed(0); torch.manual_seed(0)
    rng = np.random.RandomState(0)
    batches = [bench.synthetic_batch(rng, 2, 800, 1333) for _ in range(n_batches)]
    model, chain, opt, sync = bench.build_trainer(50, dev, 1, 2, defer=5)
    imgs = [torch.tensor(b[0], device=dev).contiguous(memory_format=torch.channels_last) for b in batches]
    losses = []
    for k in range(steps):
        b = batches[k % n_batches]
        loss = opt.update(chain, imgs[k % n_batches], b[1], b[2], b[3], b[4])
        losses.append(loss)
    opt.flush()
    torch.cuda.synchronize()
    return [float(l.item()) for l in losses]


def main():
    steps = int(os.environ.get('STEPS', 240))
    a = run(steps)
    print('loss every 20 steps:', ' '.join('%.4f' % a[i] for i in range(0, steps, 20)), '| last %.4f' % a[-1])
    print('finite:', bool(np.all(np.isfinite(a))), ' mean of first 8: %.4f  mean of last 8: %.4f'
          % (np.mean(a[:8]), np.mean(a[-8:])))
    b = run(40)
    print('bit-identical losses over 40 steps of a second run:', a[:40] == b)


if __name__ == '__main__':
    main()
