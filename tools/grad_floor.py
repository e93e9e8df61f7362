"""Developer tool: per-layer relative gradient error of the HIP train step and of a torch-CPU
fp32 run of the same graph, both against the float64 CPU graph (tests/ref_model.py) — the
measured fp32 floor that tests/test_gpu_model.py::test_train_step_matches_reference cites."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import ref_model                       # noqa: E402
from test_gpu_model import _build, _BLOCKS, _rel, H, W   # noqa: E402


def run(n_layers):
    dev = torch.device('cuda:0')
    model, chain, imgs, bboxes, labels, masks = _build(dev, n_layers)
    damp = float(os.environ.get('HEAD_DAMP', '1'))   # experiment: damp res5's residual branches
    if damp != 1:
        with torch.no_grad():
            for name, m in model.head.res5.named_modules():
                if name.endswith('bn3'):
                    m.W.mul_(damp)
    np.random.seed(123)
    loss = chain(torch.tensor(imgs, device=dev), bboxes, labels, masks, [1., 1.])
    loss.backward()
    with torch.no_grad():
        locs, scores, rois, roi_indices, anchor = model.rpn(
            model.extractor(torch.tensor(imgs, device=dev)), (H, W), [1., 1.])
    np.random.seed(123)
    ptc = chain.proposal_target_creator
    s_rois, s_idx, g_locs, g_labels, g_masks = [], [], [], [], []
    rois_h, idx_h = rois.cpu().numpy(), roi_indices.cpu().numpy()
    for i in range(2):
        a, b, c, d = ptc(rois_h[idx_h == i], bboxes[i], labels[i], masks[i])
        s_rois.append(a); s_idx.append(np.full(len(a), i, np.int32))
        g_locs.append(b); g_labels.append(c); g_masks.append(d)
    atc = chain.anchor_target_creator
    r_locs, r_labels = zip(*[atc(b, anchor.cpu().numpy(), (H, W)) for b in bboxes])
    cat = lambda xs, dt: torch.tensor(np.concatenate(xs, 0), dtype=dt)
    grads = {}
    for dtype in (torch.float64, torch.float32):
        P = ref_model.RefParams(model, dtype)
        feat = ref_model.extractor(torch.tensor(imgs), P, blocks=_BLOCKS[n_layers])
        rl, rs = ref_model.rpn(feat, P, 15)
        cls_locs, sc, mk = ref_model.head(feat, cat(s_rois, torch.float32), cat(s_idx, torch.int32), P, 81, 14)
        if dtype == torch.float64:
            print('  logits: max |score| %.3g rms %.3g; mask logits max %.3g; cls_locs max %.3g' % (
                sc.abs().max().item(), sc.pow(2).mean().sqrt().item(), mk.abs().max().item(), cls_locs.abs().max().item()))
        parts = ref_model.losses(rl, rs, cat(r_locs, torch.float32), cat(r_labels, torch.int32),
                                 cls_locs, sc, mk, cat(g_locs, torch.float32),
                                 cat(g_labels, torch.int32), cat(g_masks, torch.int32))
        sum(parts).backward()
        grads[dtype] = {n: P[n].grad for n, _ in model.named_parameters() if P[n].grad is not None}
    bad_rows = {}

    def stats(got, ref, name=None):
        got, ref = got.detach().cpu().double(), ref.detach().double()
        e2 = (got - ref).abs() / ref.abs().max().clamp_min(1e-12)
        if name is not None:
            rows = e2.reshape(e2.shape[0], -1)
            nbad = int((rows.max(1).values > 1e-4).sum())
            if nbad:
                bad_rows[name] = (nbad, rows.shape[0])
        err = e2.flatten()
        k = max(1, int(err.numel() * 0.999))
        return float(err.max()), float(err.kthvalue(k).values), float((err > 1e-4).double().mean()), \
            float(err.pow(2).mean().sqrt())
    rows = []
    for name, p in model.named_parameters():
        if name.startswith('extractor.conv1') or name.startswith('extractor.bn1') \
                or name.startswith('extractor.res2') or '.bn' in name:
            continue
        g64 = grads[torch.float64][name]
        rows.append((name,) + stats(p.grad, g64, name) + stats(grads[torch.float32][name], g64))
    rows.sort(key=lambda r: -r[3])
    print('R-%d: worst HIP max %.2e frac %.2e, worst CPU-fp32 max %.2e frac %.2e' % (
        n_layers, max(r[1] for r in rows), max(r[3] for r in rows), max(r[5] for r in rows),
        max(r[7] for r in rows)))
    print('  tensors with rows beyond 1e-4 (rows beyond / rows):', bad_rows)
    for r in rows[:3]:
        print('  %-34s hip max %.1e p99.9 %.1e frac %.1e rms %.1e | cpu32 max %.1e p99.9 %.1e frac %.1e rms %.1e' % r)


if __name__ == '__main__':
    from chainer_mask_rcnn_amd.functions import conv as C
    C.WINOGRAD_MIN_WORK = 1 << 24          # route the small test models like the full-size ones
    for use, fw in ((False, False), (True, False), (True, 'conv2d'), (True, True)):
        C.USE_WINOGRAD, C.WINOGRAD_TRAIN_FORWARD = use, fw
        print('Winograd backward:', use, ' Winograd forward in the train step:', fw)
        for n in (50, 101):
            run(n)
