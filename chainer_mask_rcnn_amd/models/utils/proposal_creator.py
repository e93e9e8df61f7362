"""ProposalCreator on the device — chainercv's proposal layer (un-vendored dependency;
ctor /root/reference/chainer_mask_rcnn/models/region_proposal_network.py:70 with the
parameters of models/mask_rcnn_resnet.py:48-52, per-image call :135-138; algorithm
SURVEY.md Appendix A.4).  Upstream copies loc/score/anchor to the host, sorts with
NumPy and scans the NMS bit-mask in Python; here every step is a HIP kernel and the
only host round trip is the final kept-count."""
import torch

from ...functions import proposal_ops as P


class ProposalCreator(object):

    def __init__(self, nms_thresh=0.7, n_train_pre_nms=12000, n_train_post_nms=2000,
                 n_test_pre_nms=6000, n_test_post_nms=300, force_cpu_nms=False, min_size=16):
        self.nms_thresh = nms_thresh
        self.n_train_pre_nms = n_train_pre_nms
        self.n_train_post_nms = n_train_post_nms
        self.n_test_pre_nms = n_test_pre_nms
        self.n_test_post_nms = n_test_post_nms
        self.force_cpu_nms = force_cpu_nms   # accepted for interface parity; no CPU path exists
        self.min_size = min_size
        self.train = True                    # chainer.config.train
        self.keep_host_copy = False          # set by MaskRCNNTrainChain (see batch())
        self.last_host_rois = None
        self.last_counts = None              # per-image proposal counts of the last batch()

    def __call__(self, loc, score, anchor, img_size, scale=1., return_indices=False):
        """loc (S,4), score (S,), anchor (S,4) device tensors -> roi (R,4) device tensor."""
        if self.train:
            n_pre, n_post = self.n_train_pre_nms, self.n_train_post_nms
        else:
            n_pre, n_post = self.n_test_pre_nms, self.n_test_post_nms
        loc = loc.detach()
        score = score.detach().reshape(-1)
        roi, valid = P.decode_clip(anchor, loc, img_size, float(self.min_size) * float(scale))
        order, n_sorted = P.topk_desc(score, n_pre if n_pre > 0 else score.numel(), valid)
        sorted_roi = P.gather_rows(roi, order, n_sorted)
        keep, n_keep = P.nms_sorted(sorted_roi, self.nms_thresh, n_sorted,
                                    limit=n_post if n_post > 0 else 0)
        nk = int(n_keep.item())              # the one host synchronisation
        keep = keep[:nk].contiguous()
        out = P.gather_rows(sorted_roi, keep)
        if return_indices:
            return out, order[keep.long()]
        return out

    def batch(self, locs, scores, anchor, img_size, scales):
        """All images of a batch at once: per-image decode / top-k / gather are queued
        without synchronising, the NMS of every image runs in ONE batched launch (one
        workgroup scans each image's mask concurrently) and the kept counts come back in a
        single host read.  Same results as calling the object once per image."""
        if self.train:
            n_pre, n_post = self.n_train_pre_nms, self.n_train_post_nms
        else:
            n_pre, n_post = self.n_test_pre_nms, self.n_test_post_nms
        n = len(locs)
        S = anchor.shape[0]
        k = min(n_pre, S) if n_pre > 0 else S
        dev = anchor.device
        sorted_rois = torch.empty((n, k, 4), dtype=torch.float32, device=dev)
        roi_all = torch.empty((n, S, 4), dtype=torch.float32, device=dev)
        valid_all = torch.empty((n, S), dtype=torch.uint8, device=dev)
        for i in range(n):
            P.decode_clip(anchor, locs[i].detach(), img_size,
                          float(self.min_size) * float(scales[i]), out=(roi_all[i], valid_all[i]))
        # every image's top-k in one set of launches
        score_all = torch.stack([scores[i].detach().reshape(-1) for i in range(n)]) \
            if not (torch.is_tensor(scores) and scores.dim() == 2) else scores.detach()
        order, counts = P.topk_desc_batched(score_all, k, valid_all)
        for i in range(n):
            sorted_rois[i] = P.gather_rows(roi_all[i], order[i], counts[i:i + 1])
        keep, n_keep = P.nms_sorted_batched(sorted_rois, counts, self.nms_thresh,
                                            limit=n_post if n_post > 0 else 0)
        if self.keep_host_copy:
            # one synchronisation brings the counts AND the data the host needs to form its
            # own copy of the proposals (the train chain samples RoIs on the host): the
            # device gather below then runs without a second device-to-host round trip
            n_keep_h, keep_h, sorted_h = n_keep.cpu(), keep.cpu(), sorted_rois.cpu()
            n_keep = n_keep_h.tolist()
            keep_np, sorted_np = keep_h.numpy(), sorted_h.numpy()
            self.last_host_rois = [sorted_np[i][keep_np[i, :n_keep[i]]] for i in range(n)]
        else:
            n_keep = n_keep.cpu().tolist()       # the one host synchronisation
            self.last_host_rois = None
        self.last_counts = list(n_keep)
        return [P.gather_rows(sorted_rois[i], keep[i, :n_keep[i]].contiguous())
                for i in range(n)]
