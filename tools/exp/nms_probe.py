"""Developer probe: the proposal chain's NMS in isolation — mask + scan kernels (in-library HIP events)
on RPN-like problems (12 000 / 6 000 score-sorted boxes, thresh 0.7, limit 2000 / 1000; 2 and 8
images) and on the per-class batch of inference (640 problems of <= 1000 boxes)."""
import os, sys, ctypes
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
import oracle
from chainer_mask_rcnn_amd import _lib
from chainer_mask_rcnn_amd.functions import proposal_ops as P

dev = torch.device('cuda:0')


def boxes(rng, n, H=800, W=1333):
    """Anchor-like boxes: centres on a stride-16 grid, sizes from the 15 anchor shapes, jittered."""
    cy, cx = rng.uniform(0, H, n), rng.uniform(0, W, n)
    s = rng.choice([32, 64, 128, 256, 512], n) * rng.uniform(0.7, 1.4, n)
    r = rng.choice([0.5, 1.0, 2.0], n)
    h, w = s * np.sqrt(r), s / np.sqrt(r)
    b = np.stack([np.clip(cy - h / 2, 0, H), np.clip(cx - w / 2, 0, W), np.clip(cy + h / 2, 0, H),
                  np.clip(cx + w / 2, 0, W)], 1).astype(np.float32)
    return b


def main():
    lib = _lib.load()
    rng = np.random.RandomState(0)
    for name, G, n, limit in (('train RPN, 2 images', 2, 12000, 2000), ('test RPN, 8 images', 8, 6000, 1000),
                              ('per-class, 640 problems', 640, 1000, 0), ('one problem', 1, 12000, 2000)):
        b = np.stack([boxes(rng, n) for _ in range(G)])
        counts = np.full((G,), n, np.int32) if G < 100 else rng.randint(0, n, G).astype(np.int32)
        bt, ct = torch.tensor(b, device=dev), torch.tensor(counts, device=dev)
        thresh = 0.7 if G < 100 else 0.5
        for _ in range(3):
            keep, nk = P.nms_sorted_batched(bt, ct, thresh, limit)
        torch.cuda.synchronize()
        lib.mrcnn_profile_enable(1)
        for _ in range(20):
            keep, nk = P.nms_sorted_batched(bt, ct, thresh, limit)
        torch.cuda.synchronize()
        prof = bench.profile_summary()
        lib.mrcnn_profile_enable(0)
        ref = oracle.nms_sorted(b[0, :counts[0]], thresh, limit if limit > 0 else -1)
        ok = np.array_equal(keep[0, :int(nk[0])].cpu().numpy(), ref)
        print('%-26s kept %5d (== oracle: %s)  mask %7.1f us  scan %7.1f us' % (
            name, int(nk[0]), ok, prof['nms_mask_kernel']['total_ms'] * 50, prof['nms_scan_kernel']['total_ms'] * 50))


if __name__ == '__main__':
    main()
