#!/usr/bin/env python
"""bench.py — images/sec of one Mask R-CNN ResNet50-C4 TRAIN STEP on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N>1: either under a launcher — python -m torch.distributed.run --nproc-per-node N ...
     bench.py --gpus N ... — or bare: bench.py then starts its own N ranks, one per GPU)

A "step" is one pass of the hot path over one synthetic batch: forward (ResNet-C4
extractor, RPN, device ProposalCreator, host target creators, ROIAlign, res5 head, five
losses) + backward + (N>1) RCCL gradient all-reduce + MomentumSGD/WeightDecay update —
the iteration of /root/reference/examples/train_common.py:226-231 on BASELINE.json
configs[1]: batch 2 x 800x1333 fp32 per GPU, 512 RoIs/img, 81 classes.  Inputs are
resident in HBM before the timed region.  Weak scaling: every rank runs its own batch of 2.

Prints ONE JSON line (rank 0) with the contract fields plus `roofline` (dominant kernel,
HIP-event timed inside the run), `rotating_h2d` (the same step over rotating host batches with
the image upload inside the timed region) and `cpu_baseline` (the oracle's restated Chainer CPU
path timed on the host cores on BASELINE configs[0] in full, reported only).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MEAN = (123.152, 115.903, 103.063)        # models/mask_rcnn_resnet.py:42
FP32_MFMA_PEAK_TFLOPS = 157.3             # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32
# split-operand kernels (the default arithmetic): six dense-bf16 MFMAs
# (2500 TFLOP/s) per fp32-equivalent multiply-add
SPLIT_MFMA_PEAK_TFLOPS = round(2500.0 / 6.0, 1)
HBM_PEAK_GBS = 8000.0
ARITHMETIC_NOTE = {
    'split_bf16x3': 'split_bf16x3: fp32 in / fp32 out / fp32 accumulate; every fp32 operand is staged as three '
                 'bf16 values whose sum is the operand exactly, six bf16 MFMAs per K step (the three '
                 'dropped cross terms: <= 2^-24 of a product each, <= 2^-23 in total); error against float64 at or below the fp32-MFMA '
                 'kernels\' (tests/test_gpu_split_bf16.py, tests/test_split_arithmetic_cpu.py)',
    'fp32': 'fp32 MFMA (v_mfma_f32_32x32x2_f32) in every GEMM kernel'}
# Algorithmic work of one train step per image: fwd 1076.6 GFLOP,
# fwd + dgrad + wgrad for every trainable layer, frozen stem/res2 forward only.
TRAIN_GFLOP_PER_IMAGE = {50: 3157.0, 101: 3644.0}


def synthetic_batch(rng, batch, H, W, n_gt=8, n_fg_class=80):
    """Deterministic COCO-shaped inputs."""
    mean = np.asarray(MEAN, np.float32)[:, None, None]
    imgs = (rng.uniform(0, 255, (batch, 3, H, W)).astype(np.float32) - mean)
    bboxes, labels, masks = [], [], []
    yy, xx = np.mgrid[0:H, 0:W]
    for _ in range(batch):
        hh = rng.uniform(32, 400, n_gt)
        ww = rng.uniform(32, 400, n_gt)
        y0 = rng.uniform(0, H - 32, n_gt)
        x0 = rng.uniform(0, W - 32, n_gt)
        b = np.stack([y0, x0, np.minimum(y0 + hh, H), np.minimum(x0 + ww, W)], 1).astype(np.float32)
        m = np.zeros((n_gt, H, W), np.int32)
        for g in range(n_gt):
            cy, cx = (b[g, 0] + b[g, 2]) / 2, (b[g, 1] + b[g, 3]) / 2
            ry, rx = (b[g, 2] - b[g, 0]) / 2, (b[g, 3] - b[g, 1]) / 2
            m[g] = (((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0)
        bboxes.append(b)
        labels.append(rng.randint(0, n_fg_class, n_gt).astype(np.int32))
        masks.append(m)
    scales = np.full((batch,), 1.6, np.float32)
    return imgs, bboxes, labels, masks, scales


def build_trainer(n_layers, device, world, lr_batch, force_dp=False, bucket_bytes=16 << 20, defer=0):
    import chainer_mask_rcnn_amd as cmr
    from chainer_mask_rcnn_amd import optimizers, parallel
    # examples/coco/train.py:36-38 + examples/train_common.py:160-169
    model = cmr.models.MaskRCNNResNet(
        n_layers=n_layers, n_fg_class=80, pretrained_model=None,
        min_size=800, max_size=1333, anchor_scales=(2, 4, 8, 16, 32), roi_size=14,
        pooling_func=cmr.functions.roi_align_2d)
    chain = cmr.models.MaskRCNNTrainChain(model).to(device)
    chain.train()
    # examples/train_common.py:176-190
    opt = optimizers.MomentumSGD(lr=0.00125 * lr_batch, momentum=0.9)
    opt.setup(chain)
    opt.add_hook(optimizers.WeightDecay(rate=0.0001))
    optimizers.disable_update(model.extractor.conv1)
    optimizers.disable_update(model.extractor.bn1)
    optimizers.disable_update(model.extractor.res2)
    from chainer_mask_rcnn_amd.links import AffineChannel2D
    for m in chain.modules():
        if isinstance(m, AffineChannel2D):
            optimizers.disable_update(m)
    stabilise_synthetic_weights(model)
    sync = None
    if world > 1 or force_dp:
        sync = parallel.DataParallelGradSync(opt, bucket_bytes=bucket_bytes)
    if defer > 0:
        # hold the weight gradients (+ update) of the first res5 block's 3x3 / 1x1 back into the
        # next step's proposal window, where the GPU is otherwise nearly idle (optimizers.py)
        a, b1, b2 = model.head.res5.a, model.head.res5.b1, model.head.res5.b2
        # (res5.a's conv1 / conv4 come last: with projected pooling their weight gradients are
        # map-sized, a tenth of the others)
        opt.defer_weight_gradients([a.conv2.W, a.conv3.W, b1.conv2.W, b1.conv1.W, b1.conv3.W,
                                    b2.conv2.W, b2.conv1.W, b2.conv3.W, a.conv1.W,
                                    a.conv4.W][:defer])
    return model, chain, opt, sync


def stabilise_synthetic_weights(model):
    """No ImageNet weights offline: keep random-init activations O(1) through the 16
    residual blocks (the real model's BN-derived affines do this) so the synthetic run
    neither overflows nor diverges.  Pure weight values — no work is skipped."""
    from chainer_mask_rcnn_amd.models.resnet_extractor import Bottleneck
    with torch.no_grad():
        model.extractor.bn1.W.fill_(1. / 64.)        # images are O(128)
        for m in model.modules():
            if isinstance(m, Bottleneck):
                m.bn3.W.fill_(0.25)
                if m.projection:
                    m.bn4.W.fill_(0.5)


def fg_saturated_sampler(base):
    """Benchmark workload device (not product code): a ProposalTargetCreator whose candidate list
    also holds jittered copies of every ground-truth box (IoU >= 0.5 with it), so that the sampler
    reaches its foreground cap round(n_sample * pos_ratio) = 128 per image
    (/root/reference/chainer_mask_rcnn/models/utils/proposal_target_creator.py:49-61,132-147) as
    it does once the RPN of a real training run proposes the objects.  A random-init RPN on
    synthetic images yields ~30 foreground RoIs per image, and the mask branch's work scales with
    that count.  The jitter comes from a private RandomState: the global np.random stream is
    consumed exactly as before."""
    cls = type(base)

    class FgSaturated(cls):
        _jitter = np.random.RandomState(12345)

        def _more(self, roi, bbox):
            bbox = np.asarray(bbox, np.float32)
            reps = int(np.ceil(1.5 * self.n_sample * self.pos_ratio / max(len(bbox), 1)))
            b = np.repeat(bbox, reps, axis=0)
            hw = np.stack([b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]], 1)
            d = self._jitter.uniform(-0.08, 0.08, (len(b), 4)).astype(np.float32)
            jit = b + d * np.concatenate([hw, hw], 1)
            roi = roi.detach().cpu().numpy() if isinstance(roi, torch.Tensor) else np.asarray(roi)
            return np.concatenate([jit.astype(np.float32), roi.astype(np.float32)], 0)

        def sample(self, roi, bbox, label, *a, **k):
            return cls.sample(self, self._more(roi, bbox), bbox, label, *a, **k)

        def __call__(self, roi, bbox, label, mask, *a, **k):
            return cls.__call__(self, self._more(roi, bbox), bbox, label, mask, *a, **k)

    obj = FgSaturated.__new__(FgSaturated)
    obj.__dict__.update(base.__dict__)
    return obj


def trained_regime_sampler(base, img_hw):
    """Benchmark workload device (not product code): `fg_saturated_sampler` whose candidate list holds,
    instead of the random-init RPN's proposals (3 - 10 feature pixels high on the synthetic batch: 94 %
    of them sample ONE point per bin), as many OBJECT-SIZED boxes — 32 .. 600 px on a side, uniformly
    placed, the generator of tools/exp/roi_align_probe.py — next to the jittered ground-truth copies:
    the foreground cap of 128 is reached AND the adaptive sampling grid of ROIAlign
    (/root/reference/chainer_mask_rcnn/functions/roi_align_2d.py:91-96: ceil(roi size / 14)) grows to
    2 x 2 .. 3 x 3 as it does on a trained model's proposals.  Private RandomState: the global np.random
    stream is consumed exactly as before."""
    sat = fg_saturated_sampler(base)
    cls = type(sat)
    H, W = img_hw

    class TrainedRegime(cls):
        _boxes = np.random.RandomState(777)

        def _more(self, roi, bbox):
            n = len(roi)
            hh, ww = self._boxes.uniform(32, 600, n), self._boxes.uniform(32, 600, n)
            y0, x0 = self._boxes.uniform(0, H - 32, n), self._boxes.uniform(0, W - 32, n)
            obj = np.stack([y0, x0, np.minimum(y0 + hh, H), np.minimum(x0 + ww, W)], 1).astype(np.float32)
            return cls._more(self, obj, bbox)

    obj = TrainedRegime.__new__(TrainedRegime)
    obj.__dict__.update(sat.__dict__)
    return obj


class SmiSampler(object):
    """Package power and shader clock sampled on a host thread while a timed region runs
    (librocm_smi64 through ctypes: rsmi_dev_power_get / rsmi_dev_gpu_clk_freq_get, one sample every
    `period` seconds (10 Hz: a handful of SMU queries per timed region, `--no-smi` turns it off — same-box A/B
    in profiles/r06_smi_ab.txt; a `rocm-smi` subprocess takes longer than the 0.5 s region).  The chip clocks
    to its power budget (MI355X_MICROARCH.md "DVFS give-back"): a throughput number without the clock
    and power it was measured at cannot be compared across boxes.  Reported only; every failure
    (library missing, sensor unsupported) yields None fields, never an exception."""

    def __init__(self, index=0, period=0.1, enabled=True):
        import threading
        self.index, self.period = index, period
        self.samples = []            # (seconds, watts or None, MHz or None)
        self._stop = threading.Event()
        self._thread = None
        self.lib = None
        self._power = self._clk = None
        if not enabled:
            return
        try:
            lib = ctypes.CDLL('librocm_smi64.so')
            # every symbol resolved HERE: a library without one of them disables the sampler
            # instead of raising inside the sampling thread
            init = getattr(lib, 'rsmi_init')
            self._power = getattr(lib, 'rsmi_dev_power_get')
            self._clk = getattr(lib, 'rsmi_dev_gpu_clk_freq_get')
            if init(ctypes.c_uint64(0)) == 0:
                self.lib = lib
        except (OSError, AttributeError):
            self.lib = None

    class _Freq(ctypes.Structure):
        _fields_ = [('has_deep_sleep', ctypes.c_bool), ('num_supported', ctypes.c_uint32),
                    ('current', ctypes.c_uint32), ('frequency', ctypes.c_uint64 * 33)]

    def read(self):
        watts = mhz = None
        if self.lib is None:
            return watts, mhz
        p, kind = ctypes.c_uint64(0), ctypes.c_int(0)
        if self._power(ctypes.c_uint32(self.index), ctypes.byref(p), ctypes.byref(kind)) == 0:
            watts = p.value / 1e6
        f = self._Freq()
        if self._clk(ctypes.c_uint32(self.index), ctypes.c_int(0), ctypes.byref(f)) == 0 \
                and f.current < 33:
            mhz = f.frequency[f.current] / 1e6
        return watts, mhz

    def _loop(self):
        t0 = time.perf_counter()
        while not self._stop.is_set():
            w, m = self.read()
            self.samples.append((time.perf_counter() - t0, w, m))
            self._stop.wait(self.period)

    def __enter__(self):
        import threading
        self.samples = []
        self._stop.clear()
        if self.lib is not None:
            self._thread = threading.Thread(target=self._loop, daemon=True)
            self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        if self._thread is not None:
            self._thread.join()
            self._thread = None
        return False

    def summary(self):
        w = [x[1] for x in self.samples if x[1] is not None]
        m = [x[2] for x in self.samples if x[2] is not None]
        return dict(power_w=round(float(np.median(w)), 1) if w else None,
                    power_w_max=round(float(max(w)), 1) if w else None,
                    sclk_mhz=round(float(np.median(m)), 0) if m else None,
                    sclk_mhz_min=round(float(min(m)), 0) if m else None,
                    samples=len(self.samples))


def profile_summary():
    from chainer_mask_rcnn_amd import _lib
    lib = _lib.load()
    out = {}
    for k in range(lib.mrcnn_profile_num_kinds()):
        ms, fl, by, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
        _lib.check(lib.mrcnn_profile_summary(k, ctypes.byref(ms), ctypes.byref(fl),
                                             ctypes.byref(by), ctypes.byref(n)), 'profile_summary')
        if n.value:
            out[lib.mrcnn_profile_kind_name(k).decode()] = dict(
                total_ms=ms.value, flops=fl.value, bytes=by.value, launches=n.value)
    return out


def roi_align_isolated(chain, device, reps=20):
    """The ROIAlign kernels ALONE on the GPU on the RoIs the last timed step sampled, in the
    configuration the head runs them in (projected pooling: the forward pools the 512-channel conv1
    map with bn1 + ReLU in its epilogue and the 2048-channel conv4 map with bn4, the pixel-owner
    backward takes the two gradients back to the map; bin stride and processing order as in the
    model's call), timed by the in-library HIP events.  Inside the step they share the GPU with
    other streams' work; this is what the kernels do with the GPU to themselves."""
    from chainer_mask_rcnn_amd import _lib
    from chainer_mask_rcnn_amd.functions import conv as C
    from chainer_mask_rcnn_amd.functions._layout import nhwc
    lib = _lib.load()
    t = chain.last_targets
    rois, idx, shape = t['sample_rois'], t['sample_roi_indices'], t['feature_shape']
    head = chain.mask_rcnn.head
    order = getattr(rois, '_mrcnn_order', None)
    r5 = torch.cat((idx.to(torch.float32)[:, None], rois), 1)[:, [0, 2, 1, 4, 3]].contiguous()
    bs = max(1, head.roi_size // 7)
    spec = C.RoiSpec(r5, head.roi_size, head.roi_size, head.spatial_scale, bin_stride=bs, order=order)
    N, _, H, W = shape
    a = head.res5.a
    maps = [(nhwc(torch.randn((N, a.conv1.W.shape[0], H, W), device=device)), a.bn1.W, a.bn1.b, True),
            (nhwc(torch.randn((N, a.conv4.W.shape[0], H, W), device=device)), a.bn4.W, a.bn4.b, False)]
    gys = None
    for it in range(reps + 3):
        if it == 3:
            torch.cuda.synchronize()
            lib.mrcnn_profile_enable(1)
        ys = [C._roi_pool_affine(z, spec, sc, sh, relu) for z, sc, sh, relu in maps]
        if gys is None:
            gys = [torch.randn_like(y) for y in ys]
        for (z, _, _, _), gy in zip(maps, gys):
            C._roi_pool_bwd(gy, spec, tuple(z.shape))
    torch.cuda.synchronize()
    prof = profile_summary()
    lib.mrcnn_profile_enable(0)
    return {k: dict(gbs=round(v['bytes'] / (v['total_ms'] * 1e-3) / 1e9, 1),
                    frac_of_hbm_peak=round(v['bytes'] / (v['total_ms'] * 1e-3) / 1e9 / HBM_PEAK_GBS, 3),
                    avg_launch_us=round(v['total_ms'] * 1e3 / v['launches'], 1))
            for k, v in prof.items() if k.startswith('roi_align')}


def pmc_traffic(kernel_name, split=False):
    """HBM-side bytes per launch of `kernel_name` from the committed rocprofv3 PMC summary
    (profiles/*_pmc_fetch_write.json: separate FETCH_SIZE and WRITE_SIZE passes over this
    same command; KB units; FETCH_SIZE doubled per MI355X_MICROARCH.md section HBM: gfx950
    tallies 128-B requests at 64 B for wide coalesced reads).  None if no summary exists."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_pmc_fetch_write.json')))
    if not files:
        return None
    m = re.match(r'conv_gemm_kernel<(\d),(\d),(\w+?)(,W8)?>', kernel_name)
    if not m:
        return None
    mode = {'FWD': 0, 'DGRAD': 1, 'WGRAD': 2}[m.group(3)]
    # profiler kinds follow the kernel symbols; the unmasked variant (<.., false, false, SPLIT, W8>)
    # is what the train step runs
    # (since round 6f the symbols carry two more arguments, PW and K3: W8 has neither; the other kinds
    # run as three symbols — the first one the summary holds stands for the kind)
    key = re.compile(r'conv_gemm_kernel<%s, %s, %d, false, false, %s, %s(, (true|false), (true|false))?>' % (
        m.group(1), m.group(2), mode, 'true' if split or m.group(4) else 'false',
        'true' if m.group(4) else 'false'))
    for path in reversed(files):          # newest summary that holds this symbol
        with open(path) as f:
            data = json.load(f)
        for k, v in data.items():
            if key.search(k) and v.get('WRITE_SIZE_KB_per_launch') is not None:
                return dict(bytes_per_launch=round((2.0 * v['FETCH_SIZE_KB_per_launch'] +
                                                    v['WRITE_SIZE_KB_per_launch']) * 1024.0),
                            source=os.path.basename(path))
    return None


def cpu_baseline():
    """The oracle ("port": oracle/np_step.py, the restated Chainer CPU path — NumPy im2col +
    BLAS convolutions, C ROIAlign, chainercv-style proposal / target creators, hand-written
    backward) timed on this node's host cores on BASELINE configs[0] IN FULL: one 800 x 1333
    image, 512 sampled RoIs, forward + backward, ONE timed iteration after a small warm-up
    iteration (BLAS thread pool, imports).  Nothing is extrapolated.  Reported only."""
    import oracle  # noqa: F401  (test infrastructure; used here only as the timed baseline)
    from oracle import np_step
    oracle.build()
    state = np.random.get_state()
    P = np_step.synthetic_params(50)
    w_in = np_step.synthetic_inputs(1, 1, 160, 224, n_gt=3, scale=1.0)
    np.random.seed(0)
    np_step.train_step(P, *w_in, n_sample=32, proposal_creator_params=dict(
        min_size=0, n_train_pre_nms=600, n_train_post_nms=100))          # warm-up (not timed)
    inputs = np_step.synthetic_inputs(0, 1, 800, 1333)
    np.random.seed(0)
    tm = {}
    t0 = time.perf_counter()
    out = np_step.train_step(P, *inputs, timings=tm)
    dt = time.perf_counter() - t0
    np.random.set_state(state)
    threads = os.cpu_count()
    try:
        from threadpoolctl import threadpool_info
        blas = [i['num_threads'] for i in threadpool_info() if i.get('user_api') == 'blas']
        if blas:
            threads = max(blas)
    except Exception:
        pass
    prev, phases = 0., {}
    for k, v in tm.items():
        phases[k] = round(v - prev, 2)
        prev = v
    return dict(value=1.0 / dt, unit='images/sec', cores=threads, kind='port', n=1,
                seconds_per_iteration=round(dt, 2),
                sample=('measured C1: BASELINE configs[0] in full — 1 x 800x1333 image, %d sampled '
                        'RoIs, forward + backward through oracle/np_step.py (NumPy im2col + BLAS on '
                        '%d threads of %d logical cores, C ROIAlign on OpenMP threads, single-threaded C NMS), 1 timed '
                        'iteration after a 160x224 warm-up iteration; loss %.4f; seconds per phase: %s'
                        % (len(out['gt_roi_labels']), threads, os.cpu_count(), out['losses']['loss'],
                           phases)))


def bench_infer(args, device, rank, steps=None, warmup=None):
    """BASELINE configs[4]: ResNet50-C4 inference, batch 8 x 1024 x 1024, 1000 proposals/img,
    per-class NMS + mask head on the <= 100 detections/img (MaskRCNN.predict minus the host
    cv2 prepare/paste, models/mask_rcnn.py:311-335).  Returns the JSON object (rank 0) or None."""
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    import chainer_mask_rcnn_amd as cmr
    from chainer_mask_rcnn_amd import _lib
    torch.manual_seed(0)
    batch = 8 if args.batch == 2 else args.batch
    H = W = 1024
    model = cmr.models.MaskRCNNResNet(
        n_layers=args.layers, n_fg_class=80, min_size=800, max_size=1333,
        anchor_scales=(2, 4, 8, 16, 32), roi_size=14).to(device)
    stabilise_synthetic_weights(model)
    with torch.no_grad():
        # random-init class scores are ~uniform (p = 1/81 < score_thresh): sharpen them so the
        # synthetic run produces detections for the NMS / mask stages to work on
        model.head.cls_loc_score.W[4 * 81:5 * 81] *= 60.
    rng = np.random.RandomState(0)
    mean = np.asarray(MEAN, np.float32)[:, None, None]
    x = torch.tensor(rng.uniform(0, 255, (batch, 3, H, W)).astype(np.float32) - mean,
                     device=device).contiguous(memory_format=torch.channels_last)
    scales = [1.6] * batch
    sizes = [(640, 640)] * batch

    def step():
        return model.predict_prepared(x, scales, sizes)

    for _ in range(warmup):
        out = step()
    torch.cuda.synchronize()
    lib = _lib.load()
    lib.mrcnn_profile_enable(1)
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    prof = profile_summary()
    lib.mrcnn_profile_enable(0)
    n_det = [len(b) for b in out[0]]
    del model, x
    if rank == 0:
        conv = {k: v for k, v in prof.items() if k.startswith('conv_gemm')}
        gflop = sum(v['flops'] for v in conv.values()) / 1e9
        ms = sum(v['total_ms'] for v in conv.values())
        from chainer_mask_rcnn_amd.functions import conv as conv_mod
        split = conv_mod.GEMM_ARITHMETIC == 'split_bf16x3' and 'split_bf16=0' not in args.tune
        infer_peak = SPLIT_MFMA_PEAK_TFLOPS if split else FP32_MFMA_PEAK_TFLOPS
        infer_note = ('split-operand arithmetic: dense bf16 MFMA 2500 TFLOP/s / 6 products per fp32 '
                      'multiply-add; achieved = nominal fp32 flops / kernel time') if split else \
            'fp32 MFMA (v_mfma_f32_32x32x2_f32)'
        return dict(
            metric='images/sec inference, ResNet%d-C4 Mask R-CNN, 8x1024x1024' % args.layers,
            value=round(steps * batch / elapsed, 3), unit='images/sec', n_gpus=1,
            steps=steps, warmup=warmup,
            ms_per_step=round(elapsed / steps * 1e3, 3), higher_is_better=True,
            scaling='weak', vs_baseline=None,
            dtype='f32 (bf16x3 split operands, f32 accumulate)' if split else 'f32',
            data='synthetic',
            config=dict(workload='BASELINE configs[4]: ResNet%d-C4 inference, batch %dx%dx%d, '
                        '1000 proposals/img, per-class NMS + mask head' % (args.layers, batch, H, W),
                        arithmetic=ARITHMETIC_NOTE['split_bf16x3' if split else 'fp32'],
                        detections_per_image=n_det,
                        executed_gemm_gflop_per_image=round(gflop / steps / batch, 1),
                        gemm_tflops=round(gflop / ms, 2)),
            roofline=dict(bound='mfma', kernel='conv_gemm_kernel (all instantiations)',
                          achieved=round(gflop / ms, 2), peak=infer_peak,
                          unit='TFLOP/s', frac=round(gflop / ms / infer_peak, 4),
                          peak_note=infer_note, traffic=None,
                          kernels={k: dict(ms_per_step=round(v['total_ms'] / steps, 3),
                                           launches_per_step=v['launches'] / steps)
                                   for k, v in prof.items()}))
    return None


def preflight_allreduce(exchange, rank, world, device, timeout_s=60.0):
    """Before any timed work of a multi-GPU run: ONE 4-float all-reduce through the gradient exchange
    (RCCL behind mrcnn_allreduce_* on an N-GPU node) with a host-side deadline.  A rank whose
    communicator did not come up, or whose peers never arrive, shows up HERE — with the rank named on
    stderr and a non-zero exit — instead of as a hang inside the first training step's bucket."""
    import threading
    t = torch.full((4,), float(rank + 1), dtype=torch.float32, device=device)
    done, err = threading.Event(), []

    def run():
        try:
            torch.cuda.set_device(device)       # (a new thread starts on device 0)
            exchange.allreduce_async(t, -3)
            exchange.wait_all()
            torch.cuda.synchronize(device)
        except Exception as e:        # noqa: BLE001  (reported below, with the rank)
            err.append(e)
        done.set()

    th = threading.Thread(target=run, daemon=True)
    th.start()
    if not done.wait(timeout_s):
        sys.stderr.write('bench.py pre-flight: rank %d of %d did not complete a 4-float all-reduce within %.0f s '
                         '(%s); peers that are missing never reached the rendezvous\n'
                         % (rank, world, timeout_s, exchange.describe()))
        sys.stderr.flush()
        os._exit(3)
    if err:
        raise SystemExit('bench.py pre-flight: rank %d of %d: all-reduce failed: %r' % (rank, world, err[0]))
    want = world * (world + 1) / 2.0
    got = t.cpu().numpy()
    if not np.all(got == want):
        raise SystemExit('bench.py pre-flight: rank %d of %d: all-reduce returned %s, expected %s (%s)'
                         % (rank, world, got.tolist(), want, exchange.describe()))
    if rank == 0:
        sys.stderr.write('bench.py pre-flight: %d-rank all-reduce ok (%s)\n' % (world, exchange.describe()))


def free_port():
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


_REAL_STDOUT = None


def emit_json(obj):
    """The ONE JSON line, written to the process's real stdout (see main())."""
    line = (json.dumps(obj) + '\n').encode()
    sys.stdout.flush()
    if _REAL_STDOUT is None:
        os.write(1, line)
    else:
        os.write(_REAL_STDOUT, line)


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start one rank per GPU ourselves
    (torch.distributed.run, rendezvous on 127.0.0.1) and pass their output through."""
    import subprocess
    n_dev = torch.cuda.device_count()
    from chainer_mask_rcnn_amd import parallel
    if n_dev < args.gpus and not (parallel.rehearsal() and n_dev >= 1):
        raise SystemExit('bench.py --gpus %d: only %d ROCm device(s) visible on this node; '
                         'one process per GPU is required (no oversubscription)'
                         % (args.gpus, n_dev))
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')     # dmabuf IPC (RCCL across processes)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
           '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1',
           '--master-port', str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--repeats', type=int, default=3,
                    help='how many times the timed region of --steps steps is run back to back; `value` '
                         'is the median region, `repeats` reports all of them')
    ap.add_argument('--layers', type=int, default=50, choices=[50, 101])
    ap.add_argument('--batch', type=int, default=2, help='images per GPU')
    ap.add_argument('--height', type=int, default=800)
    ap.add_argument('--width', type=int, default=1333)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-smi', dest='smi', action='store_false',
                    help='do not sample package power / shader clock during the timed regions')
    ap.add_argument('--no-profile', action='store_true')
    ap.add_argument('--lr-batch', type=int, default=0,
                    help='override the global batch the learning-rate rule uses (developer: '
                         'try the 8-GPU learning rate on one GPU)')
    ap.add_argument('--profile-all', action='store_true',
                    help='time every kernel kind (default: only the dominant forward-form 128x128 '
                         'GEMM, so that the timed region is barely perturbed)')
    ap.add_argument('--workload', default='train', choices=['train', 'infer'],
                    help="'train' = BASELINE configs[1..3] (the headline metric); 'infer' = "
                         "configs[4]: inference-only 8x1024x1024, 1000 proposals/img")
    ap.add_argument('--force-dp', action='store_true',
                    help='create a 1-rank RCCL communicator and run the data-parallel gradient path')
    ap.add_argument('--rotate-batches', type=int, default=4,
                    help='after the resident-batch measurement, time the same number of steps '
                         'over K pre-generated host batches with the image upload inside the '
                         'timed region (the reference converter uploads every iteration, '
                         'examples/train_common.py:219-225); 0 = skip')
    ap.add_argument('--no-fg-capped', dest='fg_capped', action='store_false',
                    help='skip the second measurement with the proposal sampler at its foreground cap')
    ap.add_argument('--no-prefetch-frozen', dest='prefetch_frozen', action='store_false',
                    help="do not run the next batch's frozen prefix (conv1 .. res2) beside the backbone "
                         'backward of the current step')
    ap.add_argument('--no-direct-head-forward', '--no-winograd-forward', dest='direct_head_forward',
                    action='store_false',
                    help="skip the measurement with res5's 3x3 forward on the direct kernel")
    ap.add_argument('--no-device-targets', dest='device_targets', action='store_false',
                    help='skip the measurement with the target creators on the device')
    ap.add_argument('--pipeline-examples', type=int, default=16,
                    help='after the other measurements, time the same number of steps fed by the train '
                         "loop's input pipeline (tools/train_loop.py) over this many synthetic decoded "
                         'examples; 0 = skip')
    ap.add_argument('--no-extra-workloads', dest='extra_workloads', action='store_false',
                    help='skip the `r101` (BASELINE configs[3] per GPU) and `infer` (configs[4]) measurements '
                         'that follow the headline on single-GPU ResNet-50 runs')
    ap.add_argument('--defer-wgrad', type=int, default=5,
                    help='number of res5 weight gradients (a.conv2, a.conv3, b1.conv2, b1.conv1, b1.conv3, ...) held back into the '
                         "next step's proposal window (single-GPU runs; 0 = off)")
    ap.add_argument('--no-fp32-mfma', '--no-split-bf16', dest='fp32_mfma', action='store_false',
                    help='skip the extra measurement on the fp32-MFMA GEMM kernels (the default arithmetic '
                         'up to round 3)')
    ap.add_argument('--arithmetic', choices=['split_bf16x3', 'fp32'], default=None,
                    help="GEMM arithmetic of the headline measurement (default: the package's default, "
                         "split_bf16x3; 'fp32' = fp32 MFMA everywhere, the headline of rounds 1-3)")
    ap.add_argument('--tune', default='',
                    help='developer: comma-separated mrcnn_set_tuning knobs, e.g. small_m_split=4')
    ap.add_argument('--bucket-mb', type=float, default=16.0,
                    help='gradient all-reduce bucket size (data-parallel runs)')
    ap.add_argument('--all-legs', action='store_true',
                    help='multi-GPU runs: also run the auxiliary single-GPU measurements (rotating_h2d, '
                         'fg_capped, device_targets, direct_head_forward, fp32_mfma, pipeline_h2d); by default '
                         'a run with --gpus > 1 times the headline region only')
    args = ap.parse_args()
    if args.gpus > 1 and not args.all_legs:
        # the scaling runs need `value` per N; every auxiliary leg is one more place where eight
        # ranks must stay in step, and none of them says anything about scaling
        args.rotate_batches = 0
        args.pipeline_examples = 0
        args.fg_capped = args.device_targets = args.direct_head_forward = args.fp32_mfma = False

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        return self_launch(args)
    # stdout carries rank 0's ONE JSON line and nothing else: native libraries (gloo's
    # connection notice, RCCL's version banner) write to fd 1 too, so fd 1 points at stderr for
    # the whole run and the JSON line goes to a saved duplicate of the real stdout
    global _REAL_STDOUT
    sys.stdout.flush()
    _REAL_STDOUT = os.dup(1)
    os.dup2(2, 1)

    from chainer_mask_rcnn_amd import parallel, _lib
    rank, world, local = parallel.init_from_env()
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    if args.force_dp and world == 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', str(free_port()))
        dist.init_process_group('gloo', rank=0, world_size=1)    # control plane (unique id store)
    # one process per GPU shares the host with its peers: keep torch's CPU thread pool (used
    # only for tiny host-side tensor plumbing) from oversubscribing the cores
    torch.set_num_threads(max(1, min(8, (os.cpu_count() or 8) // max(1, 2 * world))))
    _lib.load()                                   # fail loudly without the HIP library
    for kv in args.tune.split(','):
        if '=' in kv:
            k, v = kv.split('=')
            _lib.check(_lib.load().mrcnn_set_tuning(k.encode(), int(v)), 'set_tuning')
    if args.arithmetic is not None:
        from chainer_mask_rcnn_amd.functions import conv as _conv_mod
        _conv_mod.set_gemm_arithmetic(args.arithmetic)
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a ROCm device; there is no CPU path')
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)

    if args.workload == 'infer':
        line = bench_infer(args, device, rank)
        if line is not None:
            emit_json(line)
        return

    # examples/train_common.py:135-136 — the samplers consume the global NumPy stream
    import random
    random.seed(0)
    np.random.seed(rank)
    torch.manual_seed(0)

    rng = np.random.RandomState(rank)
    imgs, bboxes, labels, masks, scales = synthetic_batch(rng, args.batch, args.height, args.width)
    parallel_bucket_bytes = int(args.bucket_mb * 2 ** 20)
    model, chain, opt, sync = build_trainer(args.layers, device, world,
                                            args.lr_batch or args.batch * world,
                                            force_dp=args.force_dp,
                                            bucket_bytes=parallel_bucket_bytes,
                                            defer=args.defer_wgrad)
    imgs_d = torch.tensor(imgs, device=device).contiguous(memory_format=torch.channels_last)
    if sync is not None and world > 1:
        preflight_allreduce(sync.exchange, rank, world, device)

    if args.prefetch_frozen:
        chain.next_imgs = imgs_d       # resident batch: the next iteration's images are known

    def step():
        return opt.update(chain, imgs_d, bboxes, labels, masks, scales)

    def fence():
        opt.flush()                 # deferred weight gradients / updates belong to the region
        torch.cuda.synchronize()
        if sync is not None and world > 1:
            sync.exchange.barrier()               # RCCL all-reduce + device synchronise
            torch.cuda.synchronize()

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)  # control plane (gloo)
        return float(t.item())

    for _ in range(args.warmup):
        loss = step()
    fence()
    lib = _lib.load()
    if not args.no_profile:
        # mode 2: HIP events only around the forward-form 128x128 GEMM launches (the dominant
        # kernel symbol: half of the GPU time); every other kind just counts launches / flops /
        # bytes.  --profile-all times everything.
        lib.mrcnn_profile_enable(1 if args.profile_all else 2)
    if sync is not None:
        sync.exchange.timing(True)
    # The headline region, `--repeats` times back to back (each: EXACTLY args.steps steps between
    # two fences = barrier + device synchronise): the package runs these kernels at its power cap,
    # so one 0.5 s sample moves with the box and its thermal state — `value` is the MEDIAN region,
    # `repeats` keeps every region's time, and the power / shader clock sampled meanwhile go into
    # `roofline`.
    R = max(1, args.repeats)
    region_s = []
    smi = SmiSampler(local, enabled=args.smi)
    with smi:
        for _ in range(R):
            t0 = time.perf_counter()
            for _ in range(args.steps):
                loss = step()
            fence()
            region_s.append(max_over_ranks(time.perf_counter() - t0))
    smi_summary = smi.summary()
    prof = {} if args.no_profile else profile_summary()
    lib.mrcnn_profile_enable(0)
    for v in prof.values():          # accumulated over R regions -> per region of args.steps steps
        for k in ('total_ms', 'flops', 'bytes', 'launches'):
            v[k] = v[k] / R
    bucket_times = None
    if sync is not None:
        bucket_times = sync.exchange.bucket_times(len(sync.buckets.bounds))     # (None: exchange without timers)
        if bucket_times is not None:
            bucket_times = [(ms / R, by / R, n) for ms, by, n in bucket_times]
        sync.exchange.timing(False)
    n_rois = chain.last_targets['n_rois']
    loss_val = float(loss.item())
    elapsed = float(np.median(region_s))
    roi_iso = None
    if not args.no_profile and 'sample_roi_indices' in chain.last_targets:
        roi_iso = roi_align_isolated(chain, device)

    # ---- second measurement: rotating host batches, image upload inside the timed region ----
    rotating = None
    if args.rotate_batches > 0:
        K = args.rotate_batches
        host = [(imgs, bboxes, labels, masks, scales)]
        for _ in range(K - 1):
            host.append(synthetic_batch(rng, args.batch, args.height, args.width))
        pinned = [torch.from_numpy(np.ascontiguousarray(h[0])).pin_memory() for h in host]

        from chainer_mask_rcnn_amd.models.mask_rcnn_train_chain import copy_stream as _copy_stream
        copy_stream = _copy_stream(device)
        staged = {}

        def stage(k):
            """Upload batch k's images on the copy stream (25.6 MB per batch of 2)."""
            with torch.cuda.stream(copy_stream):
                t = pinned[k % K].to(device, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(copy_stream)
            staged[k] = (t, ev)

        def rot_step(k):
            # the input pipeline runs one batch ahead of the model (as a prefetching iterator
            # would): batch k was uploaded during step k-1, batch k+1 starts uploading now
            if k not in staged:
                stage(k)
            x, ev = staged.pop(k)
            torch.cuda.current_stream(device).wait_event(ev)
            x.record_stream(torch.cuda.current_stream(device))
            stage(k + 1)
            if args.prefetch_frozen:
                def nxt(k1=k + 1):
                    t, e = staged[k1]
                    torch.cuda.current_stream(device).wait_event(e)     # the upload of batch k+1
                    return t
                chain.next_imgs = nxt
            b = host[k % K]
            return opt.update(chain, x, b[1], b[2], b[3], b[4])

        for k in range(min(K, max(2, args.warmup))):
            rot_step(k)
        fence()
        t0 = time.perf_counter()
        for k in range(args.steps):
            loss_r = rot_step(k)
        fence()
        el_r = max_over_ranks(time.perf_counter() - t0)
        chain.next_imgs = imgs_d if args.prefetch_frozen else None
        rotating = dict(value=round(args.steps * args.batch * world / el_r, 3), unit='images/sec',
                        ms_per_step=round(el_r / args.steps * 1e3, 3), batches=K,
                        input='rotating+h2d: %d pre-generated pinned host batches, every step uploads '
                              'one batch (copy stream, one batch ahead of the model) inside the '
                              'timed region' % K,
                        loss=round(float(loss_r.item()), 5))

    # ---- the same step with the proposal sampler at its 128-foreground cap ----------------------
    fg_capped = None
    n_fg_default = chain.last_targets.get('n_fg')
    if args.fg_capped:
        ptc0 = chain.proposal_target_creator
        chain.proposal_target_creator = fg_saturated_sampler(ptc0)
        for _ in range(max(2, args.warmup)):
            step()
        fence()
        lib.mrcnn_profile_enable(3) if not args.no_profile else None    # count flops only
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss_c = step()
        fence()
        el_c = max_over_ranks(time.perf_counter() - t0)
        prof_c = {} if args.no_profile else profile_summary()
        lib.mrcnn_profile_enable(0)
        gf_c = sum(v['flops'] for k, v in prof_c.items() if k.startswith('conv_gemm')) / 1e9
        fg_capped = dict(value=round(args.steps * args.batch * world / el_c, 3), unit='images/sec',
                         ms_per_step=round(el_c / args.steps * 1e3, 3),
                         fg_rois_per_image=chain.last_targets['n_fg'] / float(args.batch),
                         executed_gemm_gflop_per_image=round(gf_c / args.steps / args.batch, 1) if gf_c else None,
                         workload='same step, sampler saturated: jittered copies of the ground-truth '
                                  'boxes join the proposals, so every image reaches the reference\'s '
                                  'foreground cap (128 of 512 RoIs) as in a trained run',
                         loss=round(float(loss_c.item()), 5))
        chain.proposal_target_creator = ptc0

    # ---- the regime a trained model runs in: foreground cap AND object-sized proposals ---------------
    trained = None
    if args.fg_capped:
        ptc0 = chain.proposal_target_creator
        chain.proposal_target_creator = trained_regime_sampler(ptc0, (args.height, args.width))
        for _ in range(max(2, args.warmup)):
            step()
        fence()
        lib.mrcnn_profile_enable(2) if not args.no_profile else None    # (ROIAlign launches event-timed)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss_t = step()
        fence()
        el_t = max_over_ranks(time.perf_counter() - t0)
        prof_t = {} if args.no_profile else profile_summary()
        lib.mrcnn_profile_enable(0)
        hbm_t = {k: dict(gbs=round(v['bytes'] / (v['total_ms'] * 1e-3) / 1e9, 1),
                         frac_of_hbm_peak=round(v['bytes'] / (v['total_ms'] * 1e-3) / 1e9 / HBM_PEAK_GBS, 3),
                         avg_launch_us=round(v['total_ms'] * 1e3 / v['launches'], 1))
                 for k, v in prof_t.items() if k.startswith('roi_align') and v['total_ms'] > 0}
        sr = chain.last_targets['sample_rois'].cpu().numpy()
        side = np.sqrt(np.maximum(sr[:, 2] - sr[:, 0], 1) * np.maximum(sr[:, 3] - sr[:, 1], 1))
        trained = dict(value=round(args.steps * args.batch * world / el_t, 3), unit='images/sec',
                       ms_per_step=round(el_t / args.steps * 1e3, 3),
                       fg_rois_per_image=chain.last_targets['n_fg'] / float(args.batch),
                       roi_side_px_percentiles=[round(float(v), 1) for v in np.percentile(side, [5, 50, 95])],
                       hbm_kernels=hbm_t,
                       hbm_kernels_isolated=roi_align_isolated(chain, device) if not args.no_profile else None,
                       workload='same step; the sampler sees object-sized candidate boxes (32 .. 600 px a side) '
                                'and jittered ground-truth copies: 128 foreground RoIs per image and ROIAlign '
                                'sampling grids of 2 x 2 .. 3 x 3, the regime of a trained model',
                       loss=round(float(loss_t.item()), 5))
        chain.proposal_target_creator = ptc0

    # ---- the same step with the target creators' arithmetic on the device ----------
    dev_targets = None
    if args.device_targets:
        masks_d = [torch.tensor(np.asarray(m) != 0, device=device).to(torch.uint8) for m in masks]
        chain.device_targets = True
        for _ in range(max(2, args.warmup)):
            opt.update(chain, imgs_d, bboxes, labels, masks_d, scales)
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss_d = opt.update(chain, imgs_d, bboxes, labels, masks_d, scales)
        fence()
        el_d = max_over_ranks(time.perf_counter() - t0)
        chain.device_targets = False
        dev_targets = dict(value=round(args.steps * args.batch * world / el_d, 3), unit='images/sec',
                           ms_per_step=round(el_d / args.steps * 1e3, 3),
                           workload='same step, MaskRCNNTrainChain.device_targets = True: IoU matrices, '
                                    'label rules, regression and 14x14 mask targets as HIP kernels, '
                                    'ground-truth masks resident on the device (uint8); the np.random '
                                    'draws stay on the host in the reference order (identical samples)',
                           loss=round(float(loss_d.item()), 5))

    # ---- the same step with the RoI head's 3x3 FORWARD on the direct kernel (round-2 default) ----
    wino_fwd = None
    if args.direct_head_forward:
        from chainer_mask_rcnn_amd.functions import conv as conv_mod
        prev_mode = conv_mod.WINOGRAD_TRAIN_FORWARD
        conv_mod.WINOGRAD_TRAIN_FORWARD = 'conv2d'
        for _ in range(max(2, args.warmup)):
            step()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss_w = step()
        fence()
        el_w = max_over_ranks(time.perf_counter() - t0)
        conv_mod.WINOGRAD_TRAIN_FORWARD = prev_mode
        wino_fwd = dict(value=round(args.steps * args.batch * world / el_w, 3), unit='images/sec',
                        ms_per_step=round(el_w / args.steps * 1e3, 3),
                        workload="same step, functions.conv.WINOGRAD_TRAIN_FORWARD = 'conv2d': res5's three "
                                 '3x3 forward convolutions on the direct implicit-GEMM kernel instead of the '
                                 'F(4x4,3x3) route (the default up to round 2; DESIGN.md section 4.3)',
                        loss=round(float(loss_w.item()), 5))

    # ---- the same step on the fp32-MFMA GEMM kernels (the default up to round 3) ------------------
    fp32_run = None
    from chainer_mask_rcnn_amd.functions import conv as conv_mod
    main_arithmetic = conv_mod.GEMM_ARITHMETIC
    if any(kv.split('=')[0] == 'split_bf16' for kv in args.tune.split(',') if '=' in kv):
        main_arithmetic = 'split_bf16x3' if any(
            kv.split('=')[0] == 'split_bf16' and int(kv.split('=')[1]) != 0
            for kv in args.tune.split(',') if '=' in kv) else 'fp32'
    if args.fp32_mfma and main_arithmetic != 'fp32':
        conv_mod.set_gemm_arithmetic('fp32')
        try:
            for _ in range(max(2, args.warmup)):
                step()
            fence()
            lib.mrcnn_profile_enable(2) if not args.no_profile else None
            t0 = time.perf_counter()
            for _ in range(args.steps):
                loss_s = step()
            fence()
            el_s = max_over_ranks(time.perf_counter() - t0)
            prof_f = {} if args.no_profile else profile_summary()
            lib.mrcnn_profile_enable(0)
        finally:
            conv_mod.set_gemm_arithmetic(main_arithmetic)
        fp32_run = dict(value=round(args.steps * args.batch * world / el_s, 3), unit='images/sec',
                        ms_per_step=round(el_s / args.steps * 1e3, 3),
                        workload="same step, functions.conv.set_gemm_arithmetic('fp32'): every GEMM on "
                                 'v_mfma_f32_32x32x2_f32 (157.3 TFLOP/s peak), the default arithmetic up '
                                 'to round 3',
                        loss=round(float(loss_s.item()), 5))
        conv_f = {k: v for k, v in prof_f.items() if k.startswith('conv_gemm') and v['total_ms'] > 0}
        if conv_f:
            name_f = max(conv_f, key=lambda k: conv_f[k]['total_ms'])
            ach_f = conv_f[name_f]['flops'] / (conv_f[name_f]['total_ms'] * 1e-3) / 1e12
            fp32_run['roofline'] = dict(bound='mfma', kernel=name_f, achieved=round(ach_f, 2),
                                        peak=FP32_MFMA_PEAK_TFLOPS, unit='TFLOP/s',
                                        frac=round(ach_f / FP32_MFMA_PEAK_TFLOPS, 4),
                                        avg_launch_us=round(conv_f[name_f]['total_ms'] * 1e3 /
                                                            conv_f[name_f]['launches'], 2))

    # ---- third measurement: the same step fed by the train loop's input pipeline ----------------
    pipeline = None
    if args.pipeline_examples > 0:
        sys.path.insert(0, os.path.join(ROOT, 'tools'))
        import train_loop as TL
        import chainer_mask_rcnn_amd as cmr
        data = TL.SyntheticInstances(args.pipeline_examples, seed=100 + rank,
                                     height=int(round(args.height * 0.6)), width=int(round(args.width * 0.6)),
                                     virtual_len=8192)
        it = TL.SerialIterator(TL.TransformDataset(data, cmr.datasets.MaskRCNNTransform(model)), args.batch)
        loop = TL.TrainLoop(it, chain, opt, device, prefetch_frozen=args.prefetch_frozen)
        for _ in range(max(2, args.warmup)):
            loop.step()
        fence()
        loop.host_seconds = dict(fetch=0., wait=0.)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss_p = loop.step()
        fence()
        el_p = max_over_ranks(time.perf_counter() - t0)
        loop.close()
        chain.next_imgs = None
        pipeline = dict(value=round(args.steps * args.batch * world / el_p, 3), unit='images/sec',
                        ms_per_step=round(el_p / args.steps * 1e3, 3),
                        input='tools/train_loop.py: %d decoded uint8 HWC host examples (%dx%d, 8 '
                              'instances, int32 masks) -> SerialIterator (shuffled, batch %d) -> '
                              'MaskRCNNTransform (uint8 upload, device resize + mean + flip, host box / '
                              'mask transform) -> concat_examples -> optimizer.update; the pipeline '
                              'runs one batch ahead on a worker thread and its own stream, all of it '
                              'inside the timed region'
                              % (args.pipeline_examples, data.examples[0][0].shape[0],
                                 data.examples[0][0].shape[1], args.batch),
                        worker_ms_per_batch=round(loop.host_seconds['fetch'] / args.steps * 1e3, 2),
                        worker_examples_ms_per_batch=round(loop.host_seconds.get('examples', 0.) / args.steps * 1e3, 2),
                        step_waited_ms_per_batch=round(loop.host_seconds['wait'] / args.steps * 1e3, 2),
                        loss=round(float(loss_p.detach().item()), 5))

    # ---- BASELINE configs[3] per GPU (ResNet101-C4, same batch) and configs[4] (inference) --------
    # the other two single-GPU workloads of BASELINE.json, observed by whoever runs the default command
    r101 = infer = None
    if world == 1 and args.layers == 50 and args.extra_workloads:
        model1, chain1, opt1, _ = build_trainer(101, device, 1, args.lr_batch or args.batch,
                                                defer=args.defer_wgrad)
        if args.prefetch_frozen:
            chain1.next_imgs = imgs_d
        for _ in range(max(2, args.warmup)):
            opt1.update(chain1, imgs_d, bboxes, labels, masks, scales)
        opt1.flush()
        torch.cuda.synchronize()
        lib.mrcnn_profile_enable(3) if not args.no_profile else None    # count flops only
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss_1 = opt1.update(chain1, imgs_d, bboxes, labels, masks, scales)
        opt1.flush()
        torch.cuda.synchronize()
        el_1 = time.perf_counter() - t0
        prof_1 = {} if args.no_profile else profile_summary()
        lib.mrcnn_profile_enable(0)
        gf_1 = sum(v['flops'] for k, v in prof_1.items() if k.startswith('conv_gemm')) / 1e9
        r101 = dict(value=round(args.steps * args.batch / el_1, 3), unit='images/sec',
                    ms_per_step=round(el_1 / args.steps * 1e3, 3), steps=args.steps,
                    workload='BASELINE configs[3], per-GPU part on 1 GPU: ResNet101-C4 Mask R-CNN train step '
                             '(fwd+bwd+SGD), batch %dx%dx%d fp32, %d sampled RoIs/step'
                             % (args.batch, args.height, args.width, chain1.last_targets['n_rois']),
                    reference_gflop_per_image=TRAIN_GFLOP_PER_IMAGE[101],
                    executed_gemm_gflop_per_image=round(gf_1 / args.steps / args.batch, 1) if gf_1 else None,
                    step_tflops=round(gf_1 / 1e3 / el_1, 1) if gf_1 else None,
                    loss=round(float(loss_1.item()), 5))
        del model1, chain1, opt1
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        line = bench_infer(args, device, rank, steps=max(3, args.steps // 4), warmup=2)
        infer = {k: line[k] for k in ('metric', 'value', 'unit', 'steps', 'warmup', 'ms_per_step')}
        infer['workload'] = line['config']['workload']
        infer['detections_per_image'] = line['config']['detections_per_image']
        infer['gemm_tflops'] = line['config']['gemm_tflops']
        infer['kernels'] = {k: v for k, v in sorted(line['roofline']['kernels'].items(),
                                                    key=lambda kv: -kv[1]['ms_per_step'])[:8]}

    if rank == 0:
        global_batch = args.batch * world
        value = args.steps * global_batch / elapsed
        roofline = None
        gemm_gflop = sum(v['flops'] for k, v in prof.items() if k.startswith('conv_gemm')) / 1e9
        timed = {k: v for k, v in prof.items() if v['total_ms'] > 0}
        gemm_all_timed = all(v['total_ms'] > 0 for k, v in prof.items() if k.startswith('conv_gemm'))
        gemm_ms = sum(v['total_ms'] for k, v in prof.items() if k.startswith('conv_gemm'))
        if prof:
            conv = {k: v for k, v in prof.items() if k.startswith('conv_gemm')}
            name = max(conv, key=lambda k: conv[k]['total_ms'])
            d = conv[name]
            ach = d['flops'] / (d['total_ms'] * 1e-3) / 1e12
            split_tuned = main_arithmetic == 'split_bf16x3'
            peak = SPLIT_MFMA_PEAK_TFLOPS if split_tuned else FP32_MFMA_PEAK_TFLOPS
            roofline = dict(bound='mfma', kernel=name, achieved=round(ach, 2),
                            peak=peak, unit='TFLOP/s',
                            frac=round(ach / peak, 4),
                            traffic=pmc_traffic(name, split_tuned),
                            # whole step: flops the GEMM kernels executed / step wall time / peak
                            step_frac=round(gemm_gflop / 1e3 / elapsed / peak, 4),
                            flops_counting='nominal per launch: 2*M*N*K with K = R*S*C_in (padding taps '
                                           'of the 3x3 layers counted although ~18 % of their K slices '
                                           'on 7x7 maps are skipped; Winograd launches count their '
                                           'per-frequency GEMMs)',
                            avg_launch_us=round(d['total_ms'] * 1e3 / d['launches'], 2),
                            launches_per_step=d['launches'] / args.steps,
                            # package power and shader clock while the timed regions ran (median of
                            # the samples; librocm_smi64), and the fraction against the peak at THAT
                            # clock (peak scales with the shader clock; nominal 2400 MHz)
                            power_w=smi_summary['power_w'], power_w_max=smi_summary['power_w_max'],
                            sclk_mhz=smi_summary['sclk_mhz'], sclk_mhz_min=smi_summary['sclk_mhz_min'],
                            smi_samples=smi_summary['samples'],
                            frac_at_sustained_clock=round(ach / (peak * smi_summary['sclk_mhz'] / 2400.0), 4)
                            if smi_summary['sclk_mhz'] else None,
                            # HBM-bound kernels of the step: algorithmic bytes / HIP-event time vs 8 TB/s
                            hbm_kernels={k: dict(gbs=round(v['bytes'] / (v['total_ms'] * 1e-3) / 1e9, 1),
                                                 frac_of_hbm_peak=round(v['bytes'] / (v['total_ms'] * 1e-3) / 1e9 / HBM_PEAK_GBS, 3),
                                                 avg_launch_us=round(v['total_ms'] * 1e3 / v['launches'], 1))
                                         for k, v in timed.items() if k.startswith('roi_align')},
                            # the same two kernels with the GPU to themselves (same RoIs, after the timed region)
                            hbm_kernels_isolated=roi_iso,
                            kernels={k: dict(ms_per_step=round(v['total_ms'] / args.steps, 3),
                                             tflops=round(v['flops'] / (v['total_ms'] * 1e-3) / 1e12, 2)
                                             if v['flops'] else None,
                                             gbs=round(v['bytes'] / (v['total_ms'] * 1e-3) / 1e9, 1),
                                             launches_per_step=v['launches'] / args.steps)
                                     for k, v in timed.items()})
        cfg_index = 1 if world == 1 else 2
        if args.layers == 101:
            cfg_index = 3
        config = dict(
            workload='BASELINE configs[%d]%s: ResNet%d-C4 Mask R-CNN train step '
                     '(fwd+bwd%s+SGD), batch %dx%dx%d fp32 per GPU, %d sampled RoIs/step/GPU'
                     % (cfg_index, '' if (cfg_index != 3 or world == 8) else
                        ' (per-GPU part on %d GPU%s)' % (world, 's' if world > 1 else ''),
                        args.layers, '+RCCL all-reduce' if sync is not None else '',
                        args.batch, args.height, args.width, n_rois),
            input='resident', global_batch=global_batch, rois_per_image=n_rois // args.batch,
            arithmetic=ARITHMETIC_NOTE[main_arithmetic],
            # foreground RoIs per image the sampler found on this synthetic batch (the mask branch
            # and its executed flops scale with it; `fg_capped` below pins it at the reference's cap)
            fg_rois_per_image=(n_fg_default / float(args.batch)) if n_fg_default is not None else None,
            deferred_weight_gradients=len(opt.deferred_params),
            parallelism='dp%d' % world,
            loss=round(loss_val, 5) if np.isfinite(loss_val) else None,
            # reference algorithm (mask branch on all 512 RoIs/img)
            reference_gflop_per_image=TRAIN_GFLOP_PER_IMAGE[args.layers],
            # what the GEMM kernels executed (the mask branch runs on foreground
            # RoIs only: identical loss and gradients)
            executed_gemm_gflop_per_image=round(gemm_gflop / args.steps / args.batch, 1)
            if prof else None,
            gemm_tflops=round(gemm_gflop / gemm_ms, 2) if prof and gemm_all_timed else None)
        if sync is not None:
            coll = sync.describe()
            if bucket_times is not None:
                coll['allreduce_ms_per_step'] = [round(ms / args.steps, 3) for ms, _, _ in bucket_times]
                coll['allreduce_gbs'] = [round(by / (ms * 1e-3) / 1e9, 1) if ms > 0 else None
                                         for ms, by, _ in bucket_times]
            config['collective'] = coll
        out = dict(
            metric='images/sec train step, ResNet50-C4 Mask R-CNN, COCO 800x1333'
            if args.layers == 50 else 'images/sec train step, ResNet101-C4 Mask R-CNN, COCO 800x1333',
            value=round(value, 3), unit='images/sec', n_gpus=world, steps=args.steps,
            warmup=args.warmup, ms_per_step=round(elapsed / args.steps * 1e3, 3),
            higher_is_better=True, scaling='weak', vs_baseline=None,
            dtype='f32' if main_arithmetic == 'fp32' else 'f32 (bf16x3 split operands, f32 accumulate)',
            data='synthetic', config=config, roofline=roofline,
            # every timed region (each args.steps steps between fences); `value` is the median one
            repeats=dict(n=len(region_s), value_is='median',
                         ms_per_step=[round(t / args.steps * 1e3, 3) for t in region_s],
                         median=round(float(np.median(region_s)) / args.steps * 1e3, 3),
                         min=round(min(region_s) / args.steps * 1e3, 3),
                         max=round(max(region_s) / args.steps * 1e3, 3)))
        if roofline is not None and roofline['peak'] != FP32_MFMA_PEAK_TFLOPS:
            roofline['peak_note'] = ('split-operand arithmetic: peak = dense bf16 MFMA 2500 TFLOP/s / 6 '
                                     'products per fp32 multiply-add; achieved = nominal fp32 flops / '
                                     'kernel time (the matrix pipe executes 6x as many bf16 flops: '
                                     '%.0f of 2500 TFLOP/s).  Peak at the nominal 2.4 GHz; under these '
                                     'kernels the shader clock sits at 1.75-1.94 GHz (power envelope, '
                                     'profiles/r04_clockprobe_split.txt; the fp32-MFMA kernels run at '
                                     '2.35-2.39 GHz); the matrix pipe alone sustains 270 TFLOP/s of this '
                                     'arithmetic on random operands (profiles/r04_mfma_energy_probe.txt)'
                                     % (6 * ach))
        if parallel.rehearsal():
            out['rehearsal'] = ('MRCNN_DP_REHEARSAL=1: %d ranks SHARING one GPU, gradients over gloo — a run of '
                                'the launch path, not a measurement' % world)
        if rotating is not None:
            out['rotating_h2d'] = rotating
        if pipeline is not None:
            out['pipeline_h2d'] = pipeline
        if fg_capped is not None:
            out['fg_capped'] = fg_capped
        if trained is not None:
            out['trained_regime'] = trained
        if dev_targets is not None:
            out['device_targets'] = dev_targets
        if wino_fwd is not None:
            out['direct_head_forward'] = wino_fwd
        if fp32_run is not None:
            out['fp32_mfma'] = fp32_run
        if r101 is not None:
            out['r101'] = r101
        if infer is not None:
            out['infer'] = infer
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline()
        emit_json(out)
    if sync is not None and hasattr(sync.exchange, 'close'):
        torch.cuda.synchronize()
        sync.exchange.close()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
