"""Train-loop counterpart of the reference's examples/train_common.py:200-231 for the HIP path:

    seeds (:135-136) -> dataset -> TransformDataset(MaskRCNNTransform) (:192-195)
          -> SerialIterator(batch_size_per_gpu) (:207-210) -> converter = concat_examples(padding=0,
             indices_concat=[0, 2, 3, 4], ...) (:219-225) -> StandardUpdater: optimizer.update(model,
             *batch) (:226-231)

chainer's iterator / updater classes are third-party (not in the reference repository); what is
restated here is their contract as train_common.py uses it: `SerialIterator` draws its order from
the GLOBAL `np.random` stream (a permutation at construction, a shuffle whenever an epoch ends,
batches wrap around the epoch boundary), `TransformDataset` applies the transform at access time
(so `MaskRCNNTransform`'s flip draws from Python's `random` happen in iteration order), and the
updater calls `optimizer.update(lossfun, *converter(batch))`.

MI355X arrangement: the input pipeline runs ONE BATCH AHEAD of the model on a worker thread and its
own HIP stream — JPEG decode / mask rasterisation and the host half of the transform (boxes, masks)
on the host cores, the image as decoded (uint8 HWC) over PCIe, resize + mean + flip + zero-padded
channels-last batch assembly in device kernels on that stream; the compute stream waits on an
event.  The random streams are consumed in the reference's order: the worker is the only consumer
of `random`; the iterator's `np.random` shuffle at an epoch end is drawn where the reference's
`next(iterator)` would draw it relative to the samplers' draws, so a batch that needs the NEXT
epoch's order is simply not fetched early.  `prefetch=False` is the reference's serial order and
gives bit-identical losses (tests/test_gpu_train_loop.py).

    python tools/train_loop.py --coco-root DIR --iterations 100      # a COCO-layout directory
    python tools/train_loop.py --synthetic 64 --iterations 20        # in-memory synthetic examples
"""
import functools
import os
import random
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


class TransformDataset(object):
    """chainer.datasets.TransformDataset: ``transform(dataset[i])`` at access time."""

    def __init__(self, dataset, transform):
        self._dataset, self._transform = dataset, transform

    def __len__(self):
        return len(self._dataset)

    def __getitem__(self, i):
        return self._transform(self._dataset[i])


class SerialIterator(object):
    """chainer.iterators.SerialIterator(dataset, batch_size) with its defaults repeat=True,
    shuffle=True, split into the index decision (`next_indices`, where the global `np.random`
    stream is consumed exactly as `__next__` consumes it) and the data access."""

    def __init__(self, dataset, batch_size, shuffle=True):
        self.dataset, self.batch_size = dataset, batch_size
        self._order = np.random.permutation(len(dataset)) if shuffle else None
        self.current_position, self.epoch, self.is_new_epoch = 0, 0, False

    def crosses_epoch(self):
        """Will the next batch end an epoch (and therefore shuffle)?"""
        return self.current_position + self.batch_size >= len(self.dataset)

    def next_indices(self):
        i, N = self.current_position, len(self.dataset)
        i_end = i + self.batch_size
        order = self._order
        idx = list(range(i, min(i_end, N))) if order is None else [int(j) for j in order[i:i_end]]
        if i_end >= N:
            rest = i_end - N
            if order is not None:
                np.random.shuffle(order)
            if rest > 0:
                idx += list(range(rest)) if order is None else [int(j) for j in order[:rest]]
            self.current_position = rest
            self.epoch += 1
            self.is_new_epoch = True
        else:
            self.current_position = i_end
            self.is_new_epoch = False
        return idx

    def __next__(self):
        return [self.dataset[j] for j in self.next_indices()]

    next = __next__


def make_converter(device):
    """examples/train_common.py:219-225.  The reference also moves the boxes to the device, where
    its (cupy) target creators run; this build's creators consume boxes / labels / masks on the
    host (models/mask_rcnn_train_chain.py), so only the image batch is a device tensor."""
    from chainer_mask_rcnn_amd.datasets import concat_examples
    return functools.partial(concat_examples, device=device, padding=0,
                             indices_concat=[0, 2, 3, 4],   # img, _, labels, masks, scales
                             indices_to_device=[0])


def keep_host_buffers_mapped():
    """glibc serves every allocation above 128 KB by a fresh mmap and returns it to the kernel on
    free: the input pipeline's ~100 MB of mask arrays per batch would be page-faulted in again
    every iteration (measured: that costs more than producing their contents).  Raise the mmap /
    trim thresholds so that those blocks come from — and go back to — the process heap."""
    import ctypes
    try:
        libc = ctypes.CDLL('libc.so.6')
        M_TRIM_THRESHOLD, M_MMAP_THRESHOLD = -1, -3
        libc.mallopt(M_MMAP_THRESHOLD, 1 << 30)
        libc.mallopt(M_TRIM_THRESHOLD, 2147483647)
        return True
    except OSError:
        return False


class TrainLoop(object):
    """iterator -> converter -> optimizer.update, the input pipeline one batch ahead."""

    def __init__(self, iterator, chain, optimizer, device, prefetch=True, prefetch_frozen=True):
        self.iterator, self.chain, self.optimizer = iterator, chain, optimizer
        self.prefetch_frozen = prefetch_frozen
        self.device = torch.device(device)
        self.converter = make_converter(self.device)
        self.prefetch = prefetch and self.device.type == 'cuda'
        self.iteration = 0
        self.host_seconds = dict(fetch=0., wait=0.)
        keep_host_buffers_mapped()
        self._pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix='mrcnn-input') \
            if self.prefetch else None
        from chainer_mask_rcnn_amd.models.mask_rcnn_train_chain import copy_stream
        self._stream = copy_stream(self.device) if self.prefetch else None
        self._pending = None

    # -- input pipeline -----------------------------------------------------------------------
    def _assemble(self, indices):
        t0 = time.perf_counter()
        if self._stream is None:
            batch = self.converter([self.iterator.dataset[j] for j in indices])
            self.host_seconds['fetch'] += time.perf_counter() - t0
            return batch, None
        torch.cuda.set_device(self.device)
        with torch.cuda.stream(self._stream):
            examples = [self.iterator.dataset[j] for j in indices]
            t1 = time.perf_counter()
            batch = self.converter(examples)
            ready = torch.cuda.Event()
            ready.record(self._stream)
        self.host_seconds['fetch'] += time.perf_counter() - t0
        self.host_seconds['examples'] = self.host_seconds.get('examples', 0.) + t1 - t0
        return batch, ready

    def _submit(self):
        # a batch that ends an epoch shuffles the order from the global np.random stream: that
        # draw belongs AFTER the samplers' draws of the step before it, so it is not fetched early
        if self._pool is None or self.iterator.crosses_epoch():
            return None
        return self._pool.submit(self._assemble, self.iterator.next_indices())

    def _take(self):
        t0 = time.perf_counter()
        fut, self._pending = self._pending, None
        batch, ready = fut.result() if fut is not None else self._assemble(self.iterator.next_indices())
        self.host_seconds['wait'] += time.perf_counter() - t0
        if ready is not None:
            main = torch.cuda.current_stream(self.device)
            main.wait_event(ready)
            batch[0].record_stream(main)
        return batch

    def _peek_next_images(self):
        fut = self._pending
        if fut is None or not fut.done():
            return None
        batch, ready = fut.result()
        if ready is not None:
            torch.cuda.current_stream(self.device).wait_event(ready)
        return batch[0]

    # -- updater ------------------------------------------------------------------------------
    def step(self):
        """One iteration of StandardUpdater.update_core: returns the loss (device tensor)."""
        batch = self._take()
        self._pending = self._submit()          # batch k+1 is prepared while step k runs
        if self.prefetch_frozen and hasattr(self.chain, 'next_imgs'):
            # the extractor's frozen prefix of batch k+1 runs beside step k's backbone backward,
            # if the worker has delivered that batch by then (never waits for it)
            self.chain.next_imgs = self._peek_next_images
        imgs, bboxes, labels, masks, scales = batch
        loss = self.optimizer.update(self.chain, imgs, bboxes, labels, masks, scales)
        self.iteration += 1
        return loss

    def run(self, n_iterations, report=None):
        losses = []
        for _ in range(n_iterations):
            losses.append(self.step())
            if report is not None:
                report(self, losses[-1])
        return losses

    def close(self):
        if self._pool is not None:
            if self._pending is not None:
                self._pending.result()
                self._pending = None
            self._pool.shutdown()
            self._pool = None


class SyntheticInstances(object):
    """In-memory stand-in for COCOInstanceSegmentationDataset.get_example: decoded uint8 HWC
    images with ellipse instances, (img, bboxes, labels, masks) in the dataset's dtypes.  Source
    images are 480 x 800, which MaskRCNN.prepare scales to 800 x 1333 (scale 5/3)."""

    def __init__(self, n, seed=0, height=480, width=800, n_gt=8, n_fg_class=80, virtual_len=None):
        # ``virtual_len``: report that many examples (index modulo n) — an epoch as long as a real
        # dataset's with a handful of distinct images in memory
        self.virtual_len = virtual_len
        rng = np.random.RandomState(seed)
        yy, xx = np.mgrid[0:height, 0:width]
        self.examples = []
        for _ in range(n):
            img = rng.randint(0, 256, (height, width, 3)).astype(np.uint8)
            hh, ww = rng.uniform(20, 240, n_gt), rng.uniform(20, 240, n_gt)
            y0, x0 = rng.uniform(0, height - 20, n_gt), rng.uniform(0, width - 20, n_gt)
            b = np.stack([y0, x0, np.minimum(y0 + hh, height), np.minimum(x0 + ww, width)], 1)
            masks = np.zeros((n_gt, height, width), np.int32)
            bboxes = np.zeros((n_gt, 4), np.float32)
            for g in range(n_gt):
                cy, cx = (b[g, 0] + b[g, 2]) / 2, (b[g, 1] + b[g, 3]) / 2
                ry, rx = (b[g, 2] - b[g, 0]) / 2, (b[g, 3] - b[g, 1]) / 2
                masks[g] = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0
                ys, xs = np.nonzero(masks[g])          # utils.mask_to_bbox: tight box of the mask
                bboxes[g] = (ys.min(), xs.min(), ys.max() + 1, xs.max() + 1)
            labels = rng.randint(0, n_fg_class, n_gt).astype(np.int32)
            self.examples.append((img, bboxes, labels, masks))

    def __len__(self):
        return self.virtual_len or len(self.examples)

    def __getitem__(self, i):
        return self.examples[i % len(self.examples)]


def build(dataset, n_layers=50, device='cuda:0', batch_size=2, seed=0, defer=5, prefetch=True,
          world=1):
    """Model, optimizer and loop as examples/train_common.py:135-231 builds them (COCO settings of
    examples/coco/train.py:36-38)."""
    import bench
    import chainer_mask_rcnn_amd as cmr
    random.seed(seed)                                   # :135-136
    np.random.seed(seed)
    torch.manual_seed(seed)
    device = torch.device(device)
    model, chain, opt, sync = bench.build_trainer(n_layers, device, world, batch_size * world, defer=defer)
    train_data = TransformDataset(dataset, cmr.datasets.MaskRCNNTransform(model))
    it = SerialIterator(train_data, batch_size)
    return TrainLoop(it, chain, opt, device, prefetch=prefetch), model, chain, opt


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument('--coco-root', default=None, help='COCO-layout directory (annotations/, train2014/ ...)')
    ap.add_argument('--split', default='minival')
    ap.add_argument('--synthetic', type=int, default=0, help='use N in-memory synthetic examples')
    ap.add_argument('--iterations', type=int, default=20)
    ap.add_argument('--layers', type=int, default=50, choices=[50, 101])
    ap.add_argument('--batch', type=int, default=2)
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--no-prefetch', action='store_true')
    args = ap.parse_args()
    import chainer_mask_rcnn_amd as cmr
    if os.environ.get('TORCH_THREADS'):
        torch.set_num_threads(int(os.environ['TORCH_THREADS']))
    if args.synthetic:
        data = SyntheticInstances(args.synthetic, seed=args.seed, virtual_len=4096)
    else:
        data = cmr.datasets.COCOInstanceSegmentationDataset(args.split, root_dir=args.coco_root)
    loop, model, chain, opt = build(data, args.layers, 'cuda:0', args.batch, args.seed,
                                    prefetch=not args.no_prefetch)
    for _ in range(int(os.environ.get('WARMUP', 3))):
        loop.step()
    opt.flush()
    torch.cuda.synchronize()
    loop.host_seconds = dict(fetch=0., wait=0.)
    t0 = time.perf_counter()
    losses = loop.run(args.iterations)
    opt.flush()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    loop.close()
    print('%d iterations, %.2f ms/step, %.2f img/s, loss %.5f; input pipeline: %.1f ms/batch on the '
          'worker, the step waited %.2f ms/batch for it'
          % (args.iterations, dt / args.iterations * 1e3, args.iterations * args.batch / dt,
             float(losses[-1].detach()), loop.host_seconds['fetch'] / args.iterations * 1e3,
             loop.host_seconds['wait'] / args.iterations * 1e3))


if __name__ == '__main__':
    main()
