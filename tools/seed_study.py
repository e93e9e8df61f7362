"""Developer tool: over N seeds of the random-init test networks, how often
does the whole-graph gradient criterion of tests/test_gpu_model.py::test_train_step_matches_reference
(every tensor: >= 99.9 % of entries within 1e-4 of the tensor's scale, none beyond 2e-3, against the
float64 CPU graph) hold for
  * the HIP path with the RoI head's 3x3 forward on the direct kernel (shipped default),
  * the HIP path with that forward on the Winograd route (WINOGRAD_TRAIN_FORWARD = 'stage'),
  * torch's own CPU fp32 kernels on the same graph (the fp32 floor)?
Writes profiles/<tag>_seed_study.json when given a tag.  usage: python tools/seed_study.py [n_seeds] [tag]"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import ref_model                       # noqa: E402
import chainer_mask_rcnn_amd as cmr    # noqa: E402
from chainer_mask_rcnn_amd.functions import conv as C   # noqa: E402

H, W = 160, 224
BLOCKS = {50: (3, 4, 6), 101: (3, 4, 23)}


def build(dev, n_layers, seed):
    """tests/test_gpu_model.py::_build with the seed as a parameter (weights, image, boxes)."""
    torch.manual_seed(seed)
    model = cmr.models.MaskRCNNResNet(
        n_layers, n_fg_class=80, anchor_scales=(2, 4, 8, 16, 32), roi_size=14, min_size=H, max_size=W,
        proposal_creator_params=dict(min_size=0, n_train_pre_nms=600, n_train_post_nms=100,
                                     n_test_pre_nms=300, n_test_post_nms=50))
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, cmr.links.AffineChannel2D):
                m.W.uniform_(0.4, 0.9)
                m.b.normal_(0, 0.1)
        if n_layers == 101:
            for name, m in model.extractor.res4.named_modules():
                if name.endswith('bn3'):
                    m.W.mul_(0.5)
        model.rpn.conv1.b.normal_(0, 0.1)
        model.head.deconv6.b.normal_(0, 0.1)
    chain = cmr.models.MaskRCNNTrainChain(
        model, proposal_target_creator=cmr.models.utils.ProposalTargetCreator(n_sample=32))
    chain.to(dev).train()
    rng = np.random.RandomState(seed)
    imgs = rng.uniform(-120, 130, (2, 3, H, W)).astype(np.float32)
    bboxes, labels, masks = [], [], []
    for n_gt in (2, 1):
        y0, x0 = rng.randint(5, 60, n_gt), rng.randint(5, 90, n_gt)
        hh, ww = rng.randint(60, 95, n_gt), rng.randint(60, 125, n_gt)
        b = np.stack([y0, x0, y0 + hh, x0 + ww], 1).astype(np.float32)
        m = np.zeros((n_gt, H, W), np.int32)
        for g, (a, c, e, f) in enumerate(b.astype(int)):
            m[g, a + 5:e - 5, c + 5:f - 5] = 1
        bboxes.append(b); labels.append(rng.randint(0, 80, n_gt).astype(np.int32)); masks.append(m)
    return model, chain, imgs, bboxes, labels, masks


def criterion(got, ref):
    worst_frac, worst_max = 0., 0.
    for name, g in got.items():
        r = ref[name].detach().double()
        err = (g.detach().cpu().double() - r).abs() / r.abs().max().clamp_min(1e-12)
        worst_frac = max(worst_frac, float((err > 1e-4).double().mean()))
        worst_max = max(worst_max, float(err.max()))
    return worst_frac, worst_max, bool(worst_frac <= 1e-3 and worst_max < 2e-3)


def one(dev, n_layers, seed):
    model, chain, imgs, bboxes, labels, masks = build(dev, n_layers, seed)
    keep = [n for n, _ in model.named_parameters()
            if not (n.startswith('extractor.conv1') or n.startswith('extractor.bn1')
                    or n.startswith('extractor.res2') or '.bn' in n)]
    hip = {}
    for mode in ('conv2d', 'stage'):
        C.WINOGRAD_TRAIN_FORWARD = mode
        for p in chain.parameters():
            p.grad = None
        np.random.seed(1000 + seed)
        chain(torch.tensor(imgs, device=dev), bboxes, labels, masks, [1., 1.]).backward()
        torch.cuda.synchronize()
        hip[mode] = {n: p.grad.clone() for n, p in model.named_parameters() if n in keep}
    C.WINOGRAD_TRAIN_FORWARD = 'conv2d'
    with torch.no_grad():
        _, _, rois, roi_indices, anchor = model.rpn(
            model.extractor(torch.tensor(imgs, device=dev)), (H, W), [1., 1.])
    np.random.seed(1000 + seed)
    ptc, atc = chain.proposal_target_creator, chain.anchor_target_creator
    rois_h, idx_h = rois.cpu().numpy(), roi_indices.cpu().numpy()
    parts = [ptc(rois_h[idx_h == i], bboxes[i], labels[i], masks[i]) for i in range(2)]
    r_locs, r_labels = zip(*[atc(b, anchor.cpu().numpy(), (H, W)) for b in bboxes])
    cat = lambda xs, dt: torch.tensor(np.concatenate(xs, 0), dtype=dt)
    s_idx = [np.full(len(p[0]), i, np.int32) for i, p in enumerate(parts)]
    cpu = {}
    for dtype in (torch.float64, torch.float32):
        P = ref_model.RefParams(model, dtype)
        feat = ref_model.extractor(torch.tensor(imgs), P, blocks=BLOCKS[n_layers])
        rl, rs = ref_model.rpn(feat, P, 15)
        cls_locs, sc, mk = ref_model.head(feat, cat([p[0] for p in parts], torch.float32),
                                          cat(s_idx, torch.int32), P, 81, 14)
        losses = ref_model.losses(rl, rs, cat(r_locs, torch.float32), cat(r_labels, torch.int32),
                                  cls_locs, sc, mk, cat([p[1] for p in parts], torch.float32),
                                  cat([p[2] for p in parts], torch.int32),
                                  cat([p[3] for p in parts], torch.int32))
        sum(losses).backward()
        cpu[dtype] = {n: P[n].grad for n in keep}
    ref = cpu[torch.float64]
    return dict(direct=criterion(hip['conv2d'], ref), winograd=criterion(hip['stage'], ref),
                cpu_fp32=criterion(cpu[torch.float32], ref))


def main():
    n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    tag = sys.argv[2] if len(sys.argv) > 2 else None
    dev = torch.device('cuda:0')
    C.WINOGRAD_MIN_WORK = 1 << 24          # route the small test models like the full-size ones
    out = {}
    for n_layers in (50, 101):
        rows = []
        for seed in range(n_seeds):
            r = one(dev, n_layers, seed)
            rows.append(dict(seed=seed, **{k: dict(worst_frac=v[0], worst_max=v[1], passes=v[2])
                                          for k, v in r.items()}))
            print('R-%d seed %d: ' % (n_layers, seed) + '  '.join(
                '%s frac %.1e max %.1e %s' % (k, v[0], v[1], 'ok' if v[2] else 'FAIL') for k, v in r.items()),
                flush=True)
        out['R%d' % n_layers] = dict(
            seeds=rows,
            passes={k: sum(1 for r in rows if r[k]['passes']) for k in ('direct', 'winograd', 'cpu_fp32')},
            median_worst_frac={k: float(np.median([r[k]['worst_frac'] for r in rows]))
                               for k in ('direct', 'winograd', 'cpu_fp32')})
        print('R-%d: passes of %d seeds: %s' % (n_layers, n_seeds, out['R%d' % n_layers]['passes']))
    if tag:
        out['criterion'] = 'every trainable tensor: fraction of entries with |g - g64| > 1e-4 max|g64| <= 1e-3 and max < 2e-3'
        with open(os.path.join(ROOT, 'gpurun_out', '%s_seed_study.json' % tag), 'w') as f:
            json.dump(out, f, indent=1)


if __name__ == '__main__':
    main()
