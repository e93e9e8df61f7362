"""Developer tool: which aten ops (torch-side plumbing, not library kernels) run in a train
step and what they cost on the GPU."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from torch.profiler import profile, ProfilerActivity


def main():
    dev = torch.device('cuda:0')
    import random
    random.seed(0); np.random.seed(0); torch.manual_seed(0)
    rng = np.random.RandomState(0)
    imgs, bboxes, labels, masks, scales = bench.synthetic_batch(rng, 2, 800, 1333)
    model, chain, opt, sync = bench.build_trainer(50, dev, 1, 2)
    imgs_d = torch.tensor(imgs, device=dev).contiguous(memory_format=torch.channels_last)
    for _ in range(3):
        opt.update(chain, imgs_d, bboxes, labels, masks, scales)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True,
                 with_stack=False) as prof:
        for _ in range(2):
            opt.update(chain, imgs_d, bboxes, labels, masks, scales)
        torch.cuda.synchronize()
    rows = [e for e in prof.key_averages(group_by_input_shape=True)
            if e.key.startswith('aten::') and e.self_device_time_total > 0]
    rows.sort(key=lambda e: -e.self_device_time_total)
    print('%-28s %10s %6s  %s' % ('op', 'us/step', 'calls', 'input shapes'))
    for e in rows[:40]:
        print('%-28s %10.1f %6.1f  %s' % (e.key, e.self_device_time_total / 2., e.count / 2.,
                                         str(e.input_shapes)[:150]))
    print('aten total us/step: %.1f' % (sum(e.self_device_time_total for e in rows) / 2.))


if __name__ == '__main__':
    main()
