"""The ASSEMBLED train step at BASELINE.json's full sizes, with the shipped routing thresholds
(no WINOGRAD_MIN_WORK override): configs[1] — ResNet50-C4, batch 2 x 800 x 1333, 512 RoIs / image —
and the per-GPU part of configs[3] — ResNet101-C4, same batch.  The step is the one bench.py
times (bench.build_trainer / bench.synthetic_batch: the reference's iteration,
/root/reference/examples/train_common.py:96-104,160-190,226-231 around
/root/reference/chainer_mask_rcnn/models/mask_rcnn_train_chain.py:76-189).

At these sizes no CPU oracle finishes in test time, so the statements are the size-independent ones:
finite losses, the sampler's 2 x 512 RoIs, two runs bit-identical, and every results-identical work
reduction of DESIGN.md section 4.2 (deferred weight gradients, frozen-prefix prefetch, row-sparse RPN
backward, foreground-only mask branch) against its OFF state — bit for bit where the same sums run
in the same order, to the small-model tests' tolerance where the summation order differs.
"""
import gc

import numpy as np
import pytest
import torch

import bench
from chainer_mask_rcnn_amd.functions import conv

pytestmark = pytest.mark.gpu

H, W, BATCH = 800, 1333, 2


@pytest.fixture
def shipped(dev):
    """The `dev` fixture lowers the Winograd work threshold for the small test models; here the
    shipped value must decide the routes."""
    saved = conv.WINOGRAD_MIN_WORK
    conv.WINOGRAD_MIN_WORK = 1 << 27
    yield dev
    conv.WINOGRAD_MIN_WORK = saved
    gc.collect()
    torch.cuda.empty_cache()


@pytest.fixture(scope='module')
def batch():
    return bench.synthetic_batch(np.random.RandomState(0), BATCH, H, W)


def _run(dev, batch, n_layers, steps=3, defer=5, prefetch=True, sparse=True, fg_only=True):
    """`steps` SGD iterations from fixed seeds -> (losses, weights, momenta, last targets)."""
    import random
    random.seed(0)
    np.random.seed(0)
    torch.manual_seed(0)
    imgs, bboxes, labels, masks, scales = batch
    model, chain, opt, _ = bench.build_trainer(n_layers, dev, 1, BATCH, defer=defer)
    chain.mask_branch_fg_only = fg_only
    x = torch.tensor(imgs, device=dev).contiguous(memory_format=torch.channels_last)
    chain.next_imgs = x if prefetch else None
    saved = conv.SPARSE_CONV_BACKWARD
    conv.SPARSE_CONV_BACKWARD = sparse
    try:
        losses = [float(opt.update(chain, x, bboxes, labels, masks, scales).detach()) for _ in range(steps)]
        opt.flush()
        torch.cuda.synchronize()
    finally:
        conv.SPARSE_CONV_BACKWARD = saved
    t = chain.last_targets
    out = dict(losses=losses, w=opt.arena.values.clone(), v=opt.arena.momenta.clone(),
               n_rois=int(t['n_rois']), n_fg=int(t['n_fg']),
               report={k: float(v) for k, v in chain.report.items()})
    del model, chain, opt, x
    gc.collect()
    torch.cuda.empty_cache()
    return out


def _rel(a, b):
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.mark.parametrize('n_layers', [50, 101])
def test_full_size_step_runs_and_repeats_bit_for_bit(shipped, batch, n_layers):
    """configs[1] (R-50) / per-GPU part of configs[3] (R-101): 1024 sampled RoIs, finite losses,
    the routes of the shipped policy (res5 / RPN conv1 on Winograd, 128x128 and 64x64 GEMM tiles,
    fused split-K tails), and the whole thing twice with identical bits."""
    assert conv.uses_winograd(conv.make_desc((1024, 512, 7, 7), (512, 512, 3, 3), 1, 1))
    assert not conv.uses_winograd(conv.make_desc((2, 256, 51, 84), (256, 256, 3, 3), 1, 1))
    a = _run(shipped, batch, n_layers)
    assert a['n_rois'] == BATCH * 512
    assert all(np.isfinite(a['losses'])) and all(np.isfinite(list(a['report'].values())))
    assert bool(torch.isfinite(a['w']).all()) and bool(torch.isfinite(a['v']).all())
    assert float(a['v'].abs().max()) > 0.               # the update really ran
    b = _run(shipped, batch, n_layers)
    assert a['losses'] == b['losses']
    assert torch.equal(a['w'], b['w']) and torch.equal(a['v'], b['v'])


def test_full_size_toggles_are_results_identical(shipped, batch):
    """Each work reduction against its off state at configs[1] size, R-50."""
    base = _run(shipped, batch, 50)
    # same kernels on the same inputs in the same order: bit for bit
    for name, kw in (('deferred weight gradients', dict(defer=0)),
                     ('frozen-prefix prefetch', dict(prefetch=False))):
        off = _run(shipped, batch, 50, **kw)
        assert off['losses'] == base['losses'], name
        assert torch.equal(off['w'], base['w']) and torch.equal(off['v'], base['v']), name
    # same sums in a different order (gathered rows / foreground rows only): the tolerance of the
    # small-model statements (tests/test_gpu_conv.py::test_row_sparse_3x3_backward_equals_dense,
    # tests/test_gpu_model.py::test_mask_branch_fg_only_is_results_identical)
    on = _run(shipped, batch, 50, steps=1)
    for name, kw in (('row-sparse RPN backward', dict(sparse=False)),
                     ('foreground-only mask branch', dict(fg_only=False))):
        off = _run(shipped, batch, 50, steps=1, **kw)
        assert off['n_fg'] == on['n_fg'] and off['n_rois'] == on['n_rois'], name
        for k, v in on['report'].items():
            assert abs(off['report'][k] - v) <= 1e-5 * max(abs(v), 1e-3), (name, k)
        # one SGD step from identical weights: the momenta ARE the (lr-scaled) gradients
        assert _rel(off['v'], on['v']) <= 1e-4, (name, _rel(off['v'], on['v']))
        assert _rel(off['w'], on['w']) <= 1e-6, name
