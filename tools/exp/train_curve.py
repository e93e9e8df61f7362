"""Developer check: the train step under the default arithmetic really trains — loss over a few
hundred steps on the rotating synthetic batches — and is bit-reproducible run to run."""
import os, sys, random
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench


def run(steps, n_batches=4, arithmetic=None):
    dev = torch.device('cuda:0')
    from chainer_mask_rcnn_amd.functions import conv as C
    C.set_gemm_arithmetic(arithmetic or C.DEFAULT_GEMM_ARITHMETIC)
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
    # the same run on the fp32-MFMA kernels: two fp32-class arithmetics follow the same curve (the
    # trajectories separate step by step as ReLU decisions near zero fall differently — they are two
    # valid fp32 runs of a chaotic system — but stay statistically indistinguishable)
    f = run(steps, arithmetic='fp32')
    from chainer_mask_rcnn_amd.functions import conv as C
    C.set_gemm_arithmetic(C.DEFAULT_GEMM_ARITHMETIC)
    print('fp32 MFMA, loss every 20 steps:', ' '.join('%.4f' % f[i] for i in range(0, steps, 20)), '| last %.4f' % f[-1])
    d = np.abs(np.array(a) - np.array(f)) / np.abs(np.array(f))
    print('relative difference of the two loss curves: step 0 %.2e, step 1 %.2e, step 5 %.2e, median over all steps %.2e, max %.2e'
          % (d[0], d[1], d[5], np.median(d), d.max()))
    print('mean of last 40: split %.4f  fp32 MFMA %.4f' % (np.mean(a[-40:]), np.mean(f[-40:])))


if __name__ == '__main__':
    main()
