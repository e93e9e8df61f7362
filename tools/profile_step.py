"""Developer tool: wall-clock breakdown of the host-side sections of one train step."""
import sys, os, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from chainer_mask_rcnn_amd.models.utils import ProposalTargetCreator, AnchorTargetCreator


def main():
    dev = torch.device('cuda:0')
    np.random.seed(0)
    rng = np.random.RandomState(0)
    imgs, bboxes, labels, masks, scales = bench.synthetic_batch(rng, 2, 800, 1333)
    model, chain, opt, _ = bench.build_trainer(50, dev, 1, 2)
    x = torch.tensor(imgs, device=dev).contiguous(memory_format=torch.channels_last)
    for _ in range(2):
        opt.update(chain, x, bboxes, labels, masks, scales)
    torch.cuda.synchronize()

    def T(name, fn, sync=True):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn()
        if sync: torch.cuda.synchronize()
        print('%-28s %8.2f ms' % (name, (time.perf_counter() - t0) * 1e3)); return r
    feats = T('extractor fwd', lambda: model.extractor(x))
    out = T('rpn fwd (+proposals)', lambda: model.rpn(feats, (800, 1333), scales))
    rpn_locs, rpn_scores, rois, roi_indices, anchor = out
    rois_h = T('rois -> host', lambda: (rois.cpu().numpy(), roi_indices.cpu().numpy()))
    ptc, atc = ProposalTargetCreator(), AnchorTargetCreator()
    def run_ptc():
        return [ptc(rois_h[0][rois_h[1] == i], bboxes[i], labels[i], masks[i]) for i in range(2)]
    res = T('PTC x2 (host)', run_ptc, sync=False)
    anchor_h = model.rpn.host_anchor(feats.shape[2], feats.shape[3], dev)
    T('ATC x2 (host)', lambda: [atc(b, anchor_h, (800, 1333)) for b in bboxes], sync=False)
    sr = torch.tensor(np.concatenate([r[0] for r in res]), device=dev)
    si = torch.tensor(np.concatenate([np.full(len(r[0]), i, np.int32) for i, r in enumerate(res)]), device=dev)
    T('head fwd', lambda: model.head(feats, sr, si))
    T('full step', lambda: opt.update(chain, x, bboxes, labels, masks, scales))
    T('full step', lambda: opt.update(chain, x, bboxes, labels, masks, scales))
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    opt.update(chain, x, bboxes, labels, masks, scales); torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats('cumulative').print_stats(35)


if __name__ == '__main__':
    main()
