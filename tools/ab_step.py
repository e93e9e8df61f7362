"""Developer tool: A/B a switch on the same box, alternating blocks of train steps.
usage: python tools/ab_step.py <toggle> [rounds] [steps]   toggles: fanout, stage, upload"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import chainer_mask_rcnn_amd.functions as F
from chainer_mask_rcnn_amd.models import resnet_extractor, mask_rcnn_train_chain


def toggles(name):
    if name == 'fanout':
        orig = F.fanout_rows
        plain = lambda x, rows: (x, x.index_select(0, rows))
        return (lambda: setattr(F, 'fanout_rows', orig)), (lambda: setattr(F, 'fanout_rows', plain))
    if name == 'stage':
        B = resnet_extractor.BuildingBlock
        return (lambda: setattr(B, 'fused_stage', True)), (lambda: setattr(B, 'fused_stage', False))
    if name == 'upload':
        orig = mask_rcnn_train_chain._upload
        plain = lambda a, dt, dev: torch.tensor(a, dtype=dt, device=dev)
        return (lambda: setattr(mask_rcnn_train_chain, '_upload', orig)), \
               (lambda: setattr(mask_rcnn_train_chain, '_upload', plain))
    if name == 'wside':
        import chainer_mask_rcnn_amd.functions.conv as C
        return (lambda: setattr(C, 'SMALL_WGRAD_SIDE_STREAM', True)), \
               (lambda: setattr(C, 'SMALL_WGRAD_SIDE_STREAM', False))
    if name == 'pretranspose':
        import chainer_mask_rcnn_amd.functions.conv as C
        return (lambda: setattr(C, 'PRETRANSPOSE_FILTERS', True)), \
               (lambda: setattr(C, 'PRETRANSPOSE_FILTERS', False))
    if name == 'wside_all':
        import chainer_mask_rcnn_amd.functions.conv as C
        return (lambda: setattr(C, 'SMALL_WGRAD_MAX_PIXELS', 1 << 30)), \
               (lambda: setattr(C, 'SMALL_WGRAD_MAX_PIXELS', 40000))
    if name == 'posmajor':
        from chainer_mask_rcnn_amd import _lib
        lib = _lib.load()
        return (lambda: lib.mrcnn_set_tuning(b'position_major_rows', 1)), \
               (lambda: lib.mrcnn_set_tuning(b'position_major_rows', 0))
    raise SystemExit('unknown toggle ' + name)


def main():
    name = sys.argv[1]
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    on, off = toggles(name)
    dev = torch.device('cuda:0')
    import random
    random.seed(0); np.random.seed(0); torch.manual_seed(0)
    rng = np.random.RandomState(0)
    imgs, bboxes, labels, masks, scales = bench.synthetic_batch(rng, 2, 800, 1333)
    model, chain, opt, sync = bench.build_trainer(50, dev, 1, 2)
    imgs_d = torch.tensor(imgs, device=dev).contiguous(memory_format=torch.channels_last)

    hi = torch.cuda.Stream(priority=-1) if os.environ.get('AB_HIGH_PRIO_MAIN') else None
    print('stream priority range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, 'priority_range') else None)

    def run(n):
        if hi is not None:
            hi.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(hi):
                for _ in range(n):
                    opt.update(chain, imgs_d, bboxes, labels, masks, scales)
            torch.cuda.current_stream().wait_stream(hi)
        else:
            for _ in range(n):
                opt.update(chain, imgs_d, bboxes, labels, masks, scales)
        torch.cuda.synchronize()

    res = {'on': [], 'off': []}
    for r in range(rounds):
        for label, fn in (('on', on), ('off', off)):
            fn()
            run(3)
            t0 = time.perf_counter()
            run(steps)
            res[label].append((time.perf_counter() - t0) / steps * 1e3)
    on()
    for k, v in res.items():
        print('%s %-4s ms/step: %s  mean %.2f' % (name, k, ' '.join('%.2f' % x for x in v), sum(v) / len(v)))


if __name__ == '__main__':
    main()
