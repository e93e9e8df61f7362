"""Developer probe: the RoIs one benchmark train step hands to ROIAlign — sampling-grid histogram,
taps per bin, RoIs per (row, 8-pixel tile) of the pixel-owner backward — and the forward / backward
kernels timed in isolation on exactly those RoIs (HIP events, 20 repetitions)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
import chainer_mask_rcnn_amd as cmr
import importlib
ra = importlib.import_module('chainer_mask_rcnn_amd.functions.roi_align_2d')
# this probe watches the stand-alone ROIAlign node (1024-channel map, reference order); the projected
# head pools inside its stage node: tools/exp/roi_proj_probe.py times that arrangement
importlib.import_module('chainer_mask_rcnn_amd.functions.conv').PROJECTED_POOLING = False


def main():
    dev = torch.device('cuda:0')
    import random
    random.seed(0); np.random.seed(0); torch.manual_seed(0)
    rng = np.random.RandomState(0)
    imgs, bboxes, labels, masks, scales = bench.synthetic_batch(rng, 2, 800, 1333)
    model, chain, opt, sync = bench.build_trainer(50, dev, 1, 2)
    imgs_d = torch.tensor(imgs, device=dev).contiguous(memory_format=torch.channels_last)
    seen = []
    orig = ra._ROIAlign2DFn.forward

    def fwd(ctx, x, rois, *a):
        seen.append((tuple(x.shape), rois.detach().cpu().numpy().copy(), a))
        return orig(ctx, x, rois, *a)
    ra._ROIAlign2DFn.forward = staticmethod(fwd)
    for _ in range(int(os.environ.get('PROBE_STEPS', '3'))):
        opt.update(chain, imgs_d, bboxes, labels, masks, scales)
    torch.cuda.synchronize()
    ra._ROIAlign2DFn.forward = staticmethod(orig)
    shape, rois, a = seen[-1]
    outh, outw, scale, sr, bs = a[:5]
    N, C, H, W = shape
    print('x', shape, 'rois', rois.shape, 'args', a)
    rw = np.maximum((rois[:, 3] - rois[:, 1]) * scale, 1.0)
    rh = np.maximum((rois[:, 4] - rois[:, 2]) * scale, 1.0)
    gh = np.ceil(rh / outh).astype(int) if sr == 0 else np.full(len(rois), sr)
    gw = np.ceil(rw / outw).astype(int) if sr == 0 else np.full(len(rois), sr)
    import collections
    print('grid (gh, gw) histogram:', sorted(collections.Counter(zip(gh.tolist(), gw.tolist())).items()))
    print('mean samples / bin %.2f -> taps / bin %.2f' % ((gh * gw).mean(), 4 * (gh * gw).mean()))
    print('roi h (feature px) percentiles', np.percentile(rh, [5, 25, 50, 75, 95]).round(1),
          'w', np.percentile(rw, [5, 25, 50, 75, 95]).round(1))
    # pixel-owner backward: RoIs per (image, row, 8-px tile)
    cnt = np.zeros((N, H, (W + 7) // 8), int)
    for r in rois:
        n = int(r[0])
        x0, y0, x1, y1 = r[1] * scale, r[2] * scale, r[3] * scale, r[4] * scale
        ylo, yhi = max(0, int(np.floor(max(y0, 0))) - 1), min(H - 1, int(np.floor(max(y0 + max(y1 - y0, 1), 0))) + 2)
        xlo, xhi = max(0, int(np.floor(max(x0, 0))) - 1), min(W - 1, int(np.floor(max(x0 + max(x1 - x0, 1), 0))) + 2)
        cnt[n, ylo:yhi + 1, xlo // 8:xhi // 8 + 1] += 1
    print('RoIs per (row, tile): mean %.1f max %d;  per row (any tile): max %d' % (
        cnt.mean(), cnt.max(), cnt.max(axis=2).max()))
    x = torch.randn(N, C, H, W, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    # processing order experiment: RoIs sorted by (image, 6-row band, x centre)
    yc = (rois[:, 2] + rois[:, 4]) * 0.5 * scale
    xc = (rois[:, 1] + rois[:, 3]) * 0.5 * scale
    order = np.lexsort((xc, (yc // 6).astype(int), rois[:, 0]))
    rb = np.random.RandomState(5)
    big = rois.copy()
    hh, ww = rb.uniform(32, 600, len(big)), rb.uniform(32, 600, len(big))
    big[:, 2] = rb.uniform(0, 800 - 32, len(big)); big[:, 1] = rb.uniform(0, 1333 - 32, len(big))
    big[:, 4] = np.minimum(big[:, 2] + hh, 800); big[:, 3] = np.minimum(big[:, 1] + ww, 1333)
    so = ra.spatial_order(rois[:, [2, 1, 4, 3]], rois[:, 0], scale)
    sob = ra.spatial_order(big[:, [2, 1, 4, 3]], big[:, 0], scale)
    for tag, rr, oo in (('as sampled', rois, None), ('as sampled, order=spatial_order', rois, so),
                        ('host-sorted rows', rois[order], None),
                        ('object-sized (32..600 px)', big, None),
                        ('object-sized, order=spatial_order', big, sob)):
        rd = torch.tensor(np.ascontiguousarray(rr), device=dev)
        od = None if oo is None else torch.tensor(oo, device=dev)
        for name in ('fwd', 'bwd'):
            ts = []
            for it in range(25):
                y = ra._ROIAlign2DFn.apply(x, rd, outh, outw, scale, sr, bs, od)
                gy = torch.randn_like(y)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda._sleep(400000)      # keep the queue busy while the host enqueues
                if name == 'fwd':
                    e0.record(); y = ra._ROIAlign2DFn.apply(x, rd, outh, outw, scale, sr, bs, od); e1.record()
                else:
                    e0.record(); y.backward(gy); e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            ts = np.sort(ts[5:])
            byts = 4.0 * (y.numel() + x.numel())
            print('%s %s: median %.1f us  min %.1f us  -> %.0f GB/s algorithmic (%.3f of 8 TB/s)' % (
                tag, name, np.median(ts), ts[0], byts / np.median(ts) / 1e3, byts / np.median(ts) / 1e3 / 8000))


if __name__ == '__main__':
    main()
