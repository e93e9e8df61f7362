"""The oracle's end-to-end training iteration (oracle/np_step.py) against the committed
self-consistency fixture tests/golden/train_step.npz (last row) — guards
the fixture the GPU test compares the HIP train step with — plus independent checks of the
hand-written backward pass against torch autograd on a bottleneck stage."""
import os

import numpy as np
import torch
import torch.nn.functional as F

from oracle import np_step, np_ref
from oracle.gen_golden import train_step_fixture


def test_train_step_reproduces_fixture(golden_dir):
    d = np.load(os.path.join(golden_dir, 'train_step.npz'))
    fx = train_step_fixture(np_step)
    for k in d.files:
        a, b = d[k], fx[k]
        if a.dtype.kind in 'iuUS':
            assert np.array_equal(a, b), k                       # indices, labels, masks, RNG state
        else:
            np.testing.assert_allclose(b, a, rtol=1e-4, atol=1e-7, err_msg=k)
    names = list(d['loss_names'])
    assert names == sorted(['rpn_loc_loss', 'rpn_cls_loss', 'roi_loc_loss', 'roi_cls_loss',
                            'roi_mask_loss', 'loss'])
    vals = dict(zip(names, d['loss_values']))
    assert abs(vals['loss'] - sum(v for k, v in vals.items() if k != 'loss')) < 1e-5
    assert (d['gt_roi_labels'] > 0).sum() > 0 and d['n_rois'].tolist() == [100, 100]


def test_stage_backward_matches_autograd():
    """np_step.stage_fwd / stage_bwd (BottleneckA + BottleneckB chain, hand-written reverse
    pass) vs torch autograd in float64."""
    rng = np.random.RandomState(0)
    shapes = {k: v for k, v in np_step.param_shapes(50).items() if k.startswith('extractor.res3.')}
    P = {}
    for k, shp in shapes.items():
        if k.endswith('.b'):
            P[k] = rng.standard_normal(shp) * 0.1
        elif '.bn' in k:
            P[k] = rng.uniform(0.5, 1.2, shp)
        else:
            P[k] = rng.standard_normal(shp) * np.sqrt(2. / np.prod(shp[1:]))
    x = rng.standard_normal((2, 256, 9, 11))
    y, caches = np_step.stage_fwd(x, P, 'extractor.res3', 4, 2)
    gy = rng.standard_normal(y.shape)
    G = {}
    gx = np_step.stage_bwd(gy, caches, P, 'extractor.res3', G)

    T = {k: torch.tensor(v, requires_grad=not ('.bn' in k)) for k, v in P.items()}
    xt = torch.tensor(x, requires_grad=True)

    def aff(h, pre):
        return h * T[pre + '.W'].view(1, -1, 1, 1) + T[pre + '.b'].view(1, -1, 1, 1)

    def block(h, pre, stride, proj):
        a = F.relu(aff(F.conv2d(h, T[pre + '.conv1.W'], stride=stride), pre + '.bn1'))
        a = F.relu(aff(F.conv2d(a, T[pre + '.conv2.W'], padding=1), pre + '.bn2'))
        a = aff(F.conv2d(a, T[pre + '.conv3.W']), pre + '.bn3')
        sc = aff(F.conv2d(h, T[pre + '.conv4.W'], stride=stride), pre + '.bn4') if proj else h
        return F.relu(a + sc)
    h = block(xt, 'extractor.res3.a', 2, True)
    for i in range(1, 4):
        h = block(h, 'extractor.res3.b%d' % i, 1, False)
    np.testing.assert_allclose(y, h.detach().numpy(), rtol=1e-9, atol=1e-9)
    h.backward(torch.tensor(gy))
    np.testing.assert_allclose(gx, xt.grad.numpy(), rtol=1e-8, atol=1e-9)
    for k, g in G.items():
        np.testing.assert_allclose(g, T[k].grad.numpy(), rtol=1e-8, atol=1e-9, err_msg=k)
    assert set(G) == {k for k in P if '.conv' in k}


def test_relu_decisions_can_be_recorded_and_forced():
    """np_step.RELU_PRE / RELU_FORCE (the decision-conditioned form of tests/test_gpu_step_golden.py):
    forcing the step's OWN decisions reproduces it bit for bit; flipping one unit with a non-zero
    gradient changes the gradients below it and nothing above; recording never changes a result."""
    cfg = dict(n_layers=50, H=96, W=128, batch=1, n_gt=2, n_sample=8,
               proposal_creator_params=dict(min_size=0, n_train_pre_nms=200, n_train_post_nms=40))
    P = np_step.synthetic_params(cfg['n_layers'], seed=0)
    inputs = np_step.synthetic_inputs(5, cfg['batch'], cfg['H'], cfg['W'], n_gt=cfg['n_gt'], scale=1.0)

    def run(force=None, record=None):
        np.random.seed(3)
        np_step.RELU_FORCE, np_step.RELU_PRE = force, record
        try:
            return np_step.train_step(P, *inputs, n_layers=cfg['n_layers'], n_sample=cfg['n_sample'],
                                      proposal_creator_params=cfg['proposal_creator_params'])
        finally:
            np_step.RELU_FORCE = np_step.RELU_PRE = None

    free = run()
    pre = {}
    rec = run(record=pre)
    sites = sorted(pre)
    assert 'rpn.conv1' in sites and 'head.deconv6' in sites and 'extractor.res3.a.1' in sites
    assert not any(s.startswith('extractor.res2') for s in sites)          # below freeze_at: no backward
    own = {s: p > 0 for s, p in pre.items()}
    same = run(force=own)
    for out in (rec, same):
        assert out['losses'] == free['losses']
        for k, g in free['grads'].items():
            assert np.array_equal(out['grads'][k], g), k
    # flip the decision of the res4 unit with the largest pre-activation
    site = 'extractor.res4.b3.2'
    flipped = dict(own)
    m = own[site].copy()
    idx = np.unravel_index(np.argmax(pre[site]), m.shape)
    m[idx] = False
    flipped[site] = m
    out = run(force=flipped)
    assert not np.array_equal(out['grads']['extractor.res4.b3.conv2.W'], free['grads']['extractor.res4.b3.conv2.W'])
    assert not np.array_equal(out['grads']['extractor.res3.a.conv1.W'], free['grads']['extractor.res3.a.conv1.W'])
