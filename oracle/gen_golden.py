"""Generate tests/golden/*.npz by running the REFERENCE's own code.

Runs only in the build container (needs /root/reference).  The reference's
`chainer_mask_rcnn/functions/roi_align_2d.py` and `affine_channel_2d.py` are
loaded with importlib under a stand-in `chainer` namespace (chainer itself is
not installable here; recipe: SURVEY.md Appendix D) and their CPU methods are
called directly; `_enumerate_shifted_anchor` (models/region_proposal_network.py:148-167)
and `expand_boxes` (models/mask_rcnn.py:44-60) are pure NumPy and are executed from their
own definitions.  Only inputs and outputs (data) are stored.

    python oracle/gen_golden.py
"""
import importlib.util
import os
import sys
import types

import numpy as np

REF = '/root/reference/chainer_mask_rcnn/functions'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden')


def _load_reference():
    chainer = types.ModuleType('chainer')
    cuda = types.ModuleType('chainer.cuda')
    cuda.cupy = None
    cuda.elementwise = lambda *a, **k: None
    cuda.get_array_module = lambda *a: np
    function = types.ModuleType('chainer.function')

    class Function(object):
        def retain_inputs(self, idx):
            pass
    function.Function = Function
    chainer.Function = Function
    utils = types.ModuleType('chainer.utils')
    type_check = types.ModuleType('chainer.utils.type_check')
    utils.type_check = type_check
    functions = types.ModuleType('chainer.functions')
    chainer.cuda, chainer.function, chainer.utils = cuda, function, utils
    chainer.functions = functions
    for name, m in [('chainer', chainer), ('chainer.cuda', cuda),
                    ('chainer.function', function), ('chainer.utils', utils),
                    ('chainer.utils.type_check', type_check),
                    ('chainer.functions', functions)]:
        sys.modules[name] = m
    mods = {}
    for fn in ['roi_align_2d', 'affine_channel_2d']:
        spec = importlib.util.spec_from_file_location(
            'ref_' + fn, os.path.join(REF, fn + '.py'))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        mods[fn] = m
    return mods


def _run_roi_align(mod, x, rois_xy, outh, outw, scale, sr, gy):
    f = mod.ROIAlign2D(outh, outw, scale, sr)
    y, = f.forward_cpu((x, rois_xy))
    gx, _ = f.backward_cpu((x, rois_xy), (gy,))
    return y, gx


def main():
    mods = _load_reference()
    ra = mods['roi_align_2d']
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.RandomState(1234)

    # (1) geometry of the reference's own unit test
    #     (tests/functions_tests/test_roi_align_2d.py:20-40) with random x.
    x = rng.uniform(-1, 1, (3, 3, 12, 8)).astype(np.float32)
    rois = np.array([[0, 1, 1, 6, 6], [2, 6, 2, 7, 11],
                     [1, 3, 1, 5, 10], [0, 3, 3, 3, 3]], np.float32)
    gy = rng.uniform(-1, 1, (4, 3, 5, 7)).astype(np.float32)
    for sr in (0, 1, 2):
        y, gx = _run_roi_align(ra, x, rois, 5, 7, 0.6, sr, gy)
        np.savez(os.path.join(OUT, 'roi_align_testgeom_sr%d.npz' % sr),
                 x=x, rois=rois, gy=gy, y=y, gx=gx, outh=5, outw=7,
                 spatial_scale=0.6, sampling_ratio=sr)

    # (2) the 8x8 toy map of tests/functions_tests/check_roi_align_2d.py:24-47
    toy = rng.uniform(0, 1, (1, 1, 8, 8)).astype(np.float32)
    for k, r in enumerate([[0, 0, 0, 2, 2], [0, 0, 0, 3, 2], [0, 0, 2, 6, 7]]):
        rois_k = np.array([r], np.float32)
        gy_k = np.ones((1, 1, 2, 2), np.float32)
        y, gx = _run_roi_align(ra, toy, rois_k, 2, 2, 1.0, 0, gy_k)
        np.savez(os.path.join(OUT, 'roi_align_toy%d.npz' % k),
                 x=toy, rois=rois_k, gy=gy_k, y=y, gx=gx, outh=2, outw=2,
                 spatial_scale=1.0, sampling_ratio=0)

    # (3) a C4-like case: N=2, C=8, 51x84 map, 32 proposals clipped to the
    #     800x1333 image (incl. sub-pixel and full-image boxes), 14x14,
    #     scale 1/16, adaptive sampling, boxes given yx (axes='yx' path of
    #     roi_align_2d :557-558; column swap applied here as the wrapper does).
    N, C, H, W = 2, 8, 51, 84
    x = rng.standard_normal((N, C, H, W)).astype(np.float32)
    img_h, img_w = 800., 1333.
    R = 32
    cy = rng.uniform(0, img_h, R)
    cx = rng.uniform(0, img_w, R)
    hh = np.exp(rng.uniform(np.log(4), np.log(800), R))
    ww = np.exp(rng.uniform(np.log(4), np.log(1333), R))
    b = np.stack([cy - hh / 2, cx - ww / 2, cy + hh / 2, cx + ww / 2], 1)
    b[:, 0::2] = np.clip(b[:, 0::2], 0, img_h)
    b[:, 1::2] = np.clip(b[:, 1::2], 0, img_w)
    b[0] = [0, 0, img_h, img_w]            # full image
    b[1] = [100.2, 200.7, 100.9, 201.1]    # sub-pixel
    b[2] = [790, 1300, 800, 1333]          # bottom-right corner
    idx = rng.randint(0, N, R).astype(np.float32)
    rois_yx = np.concatenate([idx[:, None], b], 1).astype(np.float32)
    rois_xy = rois_yx[:, [0, 2, 1, 4, 3]]
    gy = rng.standard_normal((R, C, 14, 14)).astype(np.float32)
    y, gx = _run_roi_align(ra, x, np.ascontiguousarray(rois_xy), 14, 14,
                           1. / 16, 0, gy)
    np.savez_compressed(os.path.join(OUT, 'roi_align_c4like.npz'),
                        x=x, rois_yx=rois_yx, gy=gy, y=y, gx=gx, outh=14,
                        outw=14, spatial_scale=1. / 16, sampling_ratio=0)

    # (4) AffineChannel2D on the shapes of tests/functions_tests/test_affine_channel_2d.py:17-31
    af = mods['affine_channel_2d']
    x = rng.uniform(-1, 1, (3, 3, 12, 8)).astype(np.float32)
    Wt = rng.random_sample((1, 3, 1, 1)).astype(np.float32)
    bt = rng.random_sample((1, 3, 1, 1)).astype(np.float32)
    gy = rng.uniform(-1, 1, x.shape).astype(np.float32)
    fn = af.AffineChannel2DFunction()
    y, = fn.forward((x, Wt, bt))
    gx, gW, gb = fn.backward((x, Wt, bt), (gy,))
    np.savez(os.path.join(OUT, 'affine_channel_2d.npz'),
             x=x, W=Wt, b=bt, gy=gy, y=y, gx=gx, gW=gW, gb=gb)
    # (5) the two pure-NumPy helpers of the model files that can be executed without chainer /
    #     chainercv / cv2: their function definitions are compiled straight out of the
    #     reference sources (the modules themselves do not import here) and run on seeded
    #     inputs.  Only inputs and outputs are stored.
    import ast

    def ref_function(path, name, namespace):
        tree = ast.parse(open(path).read())
        node = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == name][0]
        code = compile(ast.Module(body=[node], type_ignores=[]), path, 'exec')
        exec(code, namespace)
        return namespace[name]

    models = os.path.join(os.path.dirname(REF), 'models')
    enum = ref_function(os.path.join(models, 'region_proposal_network.py'),
                        '_enumerate_shifted_anchor',
                        {'np': np, 'cuda': types.SimpleNamespace(get_array_module=lambda *a: np)})
    expand = ref_function(os.path.join(models, 'mask_rcnn.py'), 'expand_boxes',
                          {'np': np, 'division': None})
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
    from oracle import np_ref
    base = np_ref.generate_anchor_base(16, (0.5, 1, 2), (2, 4, 8, 16, 32))   # input data
    np.savez_compressed(os.path.join(OUT, 'shifted_anchor.npz'), anchor_base=base, feat_stride=16,
             hw=np.array([[51, 84], [65, 65], [3, 5]]),
             a0=enum(base, 16, 51, 84), a1=enum(base, 16, 65, 65), a2=enum(base, 16, 3, 5))
    boxes = (rng.uniform(0, 900, (64, 4))).astype(np.float32)
    boxes[:, 2:] += boxes[:, :2]
    boxes[0] = [3.2, 4.7, 3.9, 5.1]
    np.savez(os.path.join(OUT, 'expand_boxes.npz'), boxes=boxes, scale=16. / 14.,
             out=expand(boxes.copy(), 16. / 14.))
    # (6) ProposalTargetCreator: the reference's OWN class body
    #     (models/utils/proposal_target_creator.py:25-184: label assignment, np.random call
    #     order, mask crop / one-hot / argmax) executed on top of the oracle's restatements of
    #     the third-party helpers it imports (chainercv bbox_iou / bbox2loc, cv2.resize
    #     INTER_LINEAR) — those helpers stay "unpinned", the control flow above them is the
    #     reference's.  Scene: 5 elliptical objects, 600 proposals, global seed 7.
    from oracle import np_infer

    def cv2_resize(img, dsize):
        chans = img[..., None] if img.ndim == 2 else img
        out = np.stack([np_infer.cv_resize_linear(chans[..., c], dsize[1], dsize[0])
                        for c in range(chans.shape[-1])], axis=-1)
        return out[..., 0] if out.shape[-1] == 1 else out

    def ref_class(path, name, namespace):
        tree = ast.parse(open(path).read())
        node = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == name][0]
        exec(compile(ast.Module(body=[node], type_ignores=[]), path, 'exec'), namespace)
        return namespace[name]

    ident = lambda a: a
    PTC = ref_class(os.path.join(models, 'utils', 'proposal_target_creator.py'),
                    'ProposalTargetCreator',
                    {'np': np, 'bbox_iou': np_ref.bbox_iou, 'bbox2loc': np_ref.bbox2loc,
                     'cuda': types.SimpleNamespace(get_array_module=lambda *a: np, to_cpu=ident,
                                                   to_gpu=ident),
                     'cv2': types.SimpleNamespace(resize=cv2_resize)})
    prng = np.random.RandomState(99)
    H, W, G, R = 240, 320, 5, 600
    y0 = prng.uniform(0, H - 60, G); x0 = prng.uniform(0, W - 60, G)
    bbox = np.stack([y0, x0, np.minimum(y0 + prng.uniform(30, 120, G), H),
                     np.minimum(x0 + prng.uniform(30, 120, G), W)], 1).astype(np.float32)
    label = prng.randint(0, 80, G).astype(np.int32)
    yy, xx = np.mgrid[0:H, 0:W]
    mask = np.zeros((G, H, W), np.int32)
    for g in range(G):
        cy, cx = (bbox[g, 0] + bbox[g, 2]) / 2, (bbox[g, 1] + bbox[g, 3]) / 2
        ry, rx = (bbox[g, 2] - bbox[g, 0]) / 2, (bbox[g, 3] - bbox[g, 1]) / 2
        mask[g] = (((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0)
    jit = bbox[prng.randint(0, G, R // 2)] + prng.uniform(-15, 15, (R // 2, 4))
    ry0 = prng.uniform(0, H - 20, R - R // 2); rx0 = prng.uniform(0, W - 20, R - R // 2)
    rnd = np.stack([ry0, rx0, ry0 + prng.uniform(10, 150, len(ry0)),
                    rx0 + prng.uniform(10, 150, len(ry0))], 1)
    roi = np.concatenate([jit, rnd], 0)
    roi[:, 0::2] = np.clip(roi[:, 0::2], 0, H); roi[:, 1::2] = np.clip(roi[:, 1::2], 0, W)
    roi = roi.astype(np.float32)
    np.random.seed(7)
    s_roi, loc, lab, m = PTC(n_sample=128)(roi, bbox, label, mask)
    after = np.random.randint(0, 1 << 30)       # where the global stream stands afterwards
    np.savez_compressed(os.path.join(OUT, 'proposal_target_creator.npz'), roi=roi, bbox=bbox,
                        label=label, mask=mask.astype(np.uint8), seed=7, n_sample=128,
                        sample_roi=s_roi, gt_roi_loc=loc, gt_roi_label=lab, gt_roi_mask=m,
                        next_randint=after)
    # (7) Inference post-processing: the reference's own `MaskRCNN._suppress` / `_to_bboxes`
    #     method bodies (models/mask_rcnn.py:178-265: de-normalisation, per-class threshold and
    #     NMS loop, rounded-area filter, the argsort-vs-rank keep expression) executed on the
    #     oracle's restatements of chainercv loc2bbox / non_maximum_suppression and a NumPy
    #     softmax (helpers unpinned, control flow the reference's).  Given probabilities are
    #     stored, so consumers need no softmax of their own.
    def ref_methods(path, cls, names, namespace):
        tree = ast.parse(open(path).read())
        node = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls][0]
        body = [n for n in node.body if isinstance(n, ast.FunctionDef) and n.name in names]
        exec(compile(ast.Module(body=body, type_ignores=[]), path, 'exec'), namespace)
        return [namespace[n] for n in names]

    def softmax(x):
        e = np.exp(x - x.max(axis=1, keepdims=True))
        return types.SimpleNamespace(array=(e / e.sum(axis=1, keepdims=True)).astype(np.float32))

    ns = {'np': np, 'chainer': types.SimpleNamespace(Variable=type('Variable', (), {})),
          'F': types.SimpleNamespace(softmax=softmax),
          'cuda': types.SimpleNamespace(to_cpu=lambda a: a),
          'loc2bbox': np_ref.loc2bbox, 'non_maximum_suppression': np_ref.non_maximum_suppression}
    suppress, to_bboxes = ref_methods(os.path.join(models, 'mask_rcnn.py'), 'MaskRCNN',
                                      ['_suppress', '_to_bboxes'], ns)
    n_class = 81
    me = types.SimpleNamespace(xp=np, n_class=n_class, nms_thresh=0.5, score_thresh=0.05,
                               loc_normalize_mean=(0., 0., 0., 0.),
                               loc_normalize_std=(0.1, 0.1, 0.2, 0.2), _detections_per_im=100)
    me._suppress = lambda b, p: suppress(me, b, p)
    drng = np.random.RandomState(5)
    sizes, scales, Rn = [(600, 900), (480, 640)], [1.6, 1.25], [260, 140]
    rois, idx = [], []
    for i, (sz, sc, r) in enumerate(zip(sizes, scales, Rn)):
        yy0 = drng.uniform(0, sz[0] * sc - 20, r); xx0 = drng.uniform(0, sz[1] * sc - 20, r)
        rois.append(np.stack([yy0, xx0, np.minimum(yy0 + drng.uniform(10, 300, r), sz[0] * sc),
                              np.minimum(xx0 + drng.uniform(10, 300, r), sz[1] * sc)], 1))
        idx.append(np.full(r, i, np.int32))
    rois = np.concatenate(rois).astype(np.float32)
    idx = np.concatenate(idx)
    locs = (drng.standard_normal((len(rois), n_class * 4)) * 0.5).astype(np.float32)
    logits = (drng.standard_normal((len(rois), n_class)) * 2.5).astype(np.float32)
    probs = softmax(logits).array
    bb, ll, ss = to_bboxes(me, locs.copy(), logits.copy(), rois.copy(), idx, sizes, scales)
    np.savez_compressed(
        os.path.join(OUT, 'to_bboxes.npz'), roi_cls_locs=locs, roi_scores=logits, probs=probs,
        rois=rois, roi_indices=idx, sizes=np.array(sizes), scales=np.array(scales),
        n_det=np.array([len(b) for b in bb]), bbox=np.concatenate(bb), label=np.concatenate(ll),
        score=np.concatenate(ss))
    # (8) localisation loss: the reference's own `_smooth_l1_loss` / `_fast_rcnn_loc_loss`
    #     (models/mask_rcnn_train_chain.py:192-213) with chainer's elementwise F.absolute /
    #     F.square / F.sum mapped to the NumPy functions of the same meaning.
    class Var(np.ndarray):
        @property
        def data(self):
            return self.view(np.ndarray)

    tc = os.path.join(models, 'mask_rcnn_train_chain.py')
    lns = {'np': np, 'F': types.SimpleNamespace(absolute=np.absolute, square=np.square, sum=np.sum),
           'chainer': types.SimpleNamespace(cuda=types.SimpleNamespace(get_array_module=lambda *a: np))}
    ref_function(tc, '_smooth_l1_loss', lns)
    loc_loss = ref_function(tc, '_fast_rcnn_loc_loss', lns)
    lrng = np.random.RandomState(21)
    pred = (lrng.standard_normal((700, 4)) * 0.7).astype(np.float32)
    gt = (lrng.standard_normal((700, 4)) * 0.7).astype(np.float32)
    lab = lrng.choice(np.array([-1, 0, 0, 1, 5, 80], np.int32), 700)
    np.savez(os.path.join(OUT, 'loc_loss.npz'), pred=pred, gt=gt, label=lab,
             loss_sigma3=np.float64(loc_loss(pred.view(Var), gt, lab, 3.)),
             loss_sigma1=np.float64(loc_loss(pred.view(Var), gt, lab, 1.)))
    # (9) image I/O of predict: the reference's own `MaskRCNN.prepare` (models/mask_rcnn.py:
    #     152-176: scale rule, transposes, mean subtraction) and `segm_results` (:63-107: box
    #     expansion, int truncation, clipping and paste) executed with `cv2.resize` mapped to the
    #     oracle's INTER_LINEAR restatement (cv2 is not installable here: that call stays
    #     unpinned, everything around it is the reference's).
    def cv2_resize_api(img, dsize, fx=None, fy=None):
        chans = img[..., None] if img.ndim == 2 else img
        if dsize is None:
            out_h, out_w = int(np.round(img.shape[0] * fy)), int(np.round(img.shape[1] * fx))
            sy, sx = 1. / fy, 1. / fx
        else:
            (out_w, out_h), sy, sx = dsize, None, None
        if img.dtype == np.uint8:      # cv2.resize dispatches on depth: 8-bit fixed-point path
            out = np.stack([np_infer.cv_resize_linear_u8(np.ascontiguousarray(chans[..., c]),
                                                         int(out_h), int(out_w), sy, sx)
                            for c in range(chans.shape[-1])], axis=-1)
            return out[..., 0] if img.ndim == 2 else out
        out = np.stack([np_infer.cv_resize_linear(chans[..., c].astype(np.float32), int(out_h),
                                                  int(out_w), sy, sx)
                        for c in range(chans.shape[-1])], axis=-1)
        return out[..., 0] if img.ndim == 2 else out

    cvns = types.SimpleNamespace(resize=cv2_resize_api)
    mpath = os.path.join(models, 'mask_rcnn.py')
    prepare, = ref_methods(mpath, 'MaskRCNN', ['prepare'], {'np': np, 'cv2': cvns})
    mean = np.array([123.152, 115.903, 103.063], np.float32)[:, None, None]
    pm = types.SimpleNamespace(min_size=160, max_size=240, mean=mean)
    irng = np.random.RandomState(31)
    imgs = [irng.randint(0, 256, (3, 97, 131)).astype(np.uint8),
            irng.uniform(0, 255, (3, 120, 90)).astype(np.float32),
            irng.randint(0, 256, (3, 60, 200)).astype(np.uint8)]
    outs, psizes, pscales = prepare(pm, imgs)
    np.savez_compressed(os.path.join(OUT, 'prepare.npz'), min_size=160, max_size=240,
                        mean=mean.ravel(), img0=imgs[0], img1=imgs[1], img2=imgs[2],
                        out0=outs[0], out1=outs[1], out2=outs[2], sizes=np.array(psizes),
                        scales=np.array(pscales))

    sns = {'np': np, 'cv2': cvns}
    sns['expand_boxes'] = expand
    segm = ref_function(mpath, 'segm_results', sns)
    Dn, n_fg, M, im_h, im_w = 24, 80, 14, 150, 210
    logits = (irng.standard_normal((Dn, n_fg, M, M)) * 3).astype(np.float32)
    slabel = irng.randint(0, n_fg, Dn).astype(np.int32)
    sy0 = irng.uniform(-10, im_h - 5, Dn); sx0 = irng.uniform(-10, im_w - 5, Dn)
    sbox = np.stack([sy0, sx0, sy0 + irng.uniform(1, 120, Dn), sx0 + irng.uniform(1, 150, Dn)],
                    1).astype(np.float32)
    sbox[0] = [3.2, 4.7, 3.9, 5.1]
    prob = (1. / (1. + np.exp(-logits.astype(np.float64)))).astype(np.float32)   # F.sigmoid
    masks = segm(sbox, slabel, prob, im_h, im_w)
    # only the selected class planes are needed by a consumer
    np.savez_compressed(os.path.join(OUT, 'segm_results.npz'), bbox=sbox, label=slabel,
                        logits_sel=logits[np.arange(Dn), slabel], n_fg=n_fg, im_h=im_h, im_w=im_w,
                        masks=np.packbits(masks, axis=-1), masks_shape=np.array(masks.shape))
    # (10) end-to-end self-consistency fixture (last row): the oracle's
    #      whole training iteration (oracle/np_step.py: extractor, RPN, ProposalCreator, target
    #      creators in the reference's np.random order, ROIAlign, res5 head, five losses,
    #      hand-written backward) on one tiny seeded batch with seeded weights.  Only outputs are
    #      stored; inputs and weights are regenerated from the seeds by np_step.synthetic_*.
    from oracle import np_step
    np.savez_compressed(os.path.join(OUT, 'train_step.npz'), **train_step_fixture(np_step))
    # (11) Detectron -> chainer weight mapping: the reference converter's OWN assignment
    #      statements (examples/coco/convert_caffe2_to_chainer.py:46-249) executed on seeded
    #      synthetic blobs of the real shapes; position-sensitive checksums of every destination
    #      array are stored (the arrays themselves are 170 MB).
    np.savez_compressed(os.path.join(OUT, 'detectron_convert.npz'), **detectron_fixture())
    # (12) COCO annotation decoding: the reference's own `_annotations_to_example`
    #      (datasets/coco.py:123-176) and `utils.mask_to_bbox` (utils/geometry.py:150-166) executed
    #      on a synthetic annotation list (polygons through the real PIL.ImageDraw; run-length
    #      masks through the oracle's restatement of pycocotools.mask, which is not installable).
    np.savez_compressed(os.path.join(OUT, 'coco_example.npz'), **coco_fixture())
    print('golden vectors written to', os.path.normpath(OUT))


def coco_annotations(height=60, width=80):
    """A synthetic COCO annotation list covering every branch of _annotations_to_example."""
    from oracle import np_data
    rng = np.random.RandomState(8)
    blob = np.zeros((height, width), np.uint8)
    blob[10:30, 20:50] = 1
    blob[15:20, 25:30] = 0
    ring = np.zeros((height, width), np.uint8)
    yy, xx = np.mgrid[0:height, 0:width]
    ring[((yy - 30) ** 2 + (xx - 40) ** 2 < 400) & ((yy - 30) ** 2 + (xx - 40) ** 2 > 100)] = 1
    noise = (rng.uniform(size=(height, width)) > 0.7).astype(np.uint8)
    return [
        dict(id=1, image_id=7, category_id=18, iscrowd=0, area=300.5,
             segmentation=[[5.0, 5.0, 40.5, 8.0, 30.0, 30.25, 8.0, 25.0],
                           [50.0, 40.0, 70.0, 42.0, 60.0, 55.0]]),
        dict(id=2, image_id=7, category_id=1, iscrowd=1, area=float(blob.sum()),
             segmentation=dict(size=[height, width], counts=np_data.mask_to_rle_counts(blob))),
        dict(id=3, image_id=7, category_id=44, iscrowd=0, area=float(ring.sum()),
             segmentation=dict(size=[height, width],
                               counts=np_data.rle_to_string(np_data.mask_to_rle_counts(ring)))),
        dict(id=4, image_id=7, category_id=3, iscrowd=0, area=1.0),            # no segmentation
        dict(id=5, image_id=7, category_id=3, iscrowd=0, area=float(noise.sum()),
             segmentation=dict(size=[height, width],
                               counts=np_data.rle_to_string(np_data.mask_to_rle_counts(noise)))),
        dict(id=6, image_id=7, category_id=90, iscrowd=0, area=4.0,            # malformed size
             segmentation=dict(size=[height - 1, width], counts=[0, (height - 1) * width])),
    ]


COCO_CATEGORIES = [dict(id=i, name='cat%d' % i) for i in (90, 1, 3, 18, 44)]


def coco_fixture():
    import json
    import PIL.Image
    import PIL.ImageDraw
    from oracle import np_data
    H, W = 60, 80
    anns = coco_annotations(H, W)
    base = '/root/reference/chainer_mask_rcnn'
    import ast

    def ref_function(path, name, namespace):
        tree = ast.parse(open(path).read())
        node = [n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef) and n.name == name][0]
        exec(compile(ast.Module(body=[node], type_ignores=[]), path, 'exec'), namespace)
        return namespace[name]
    mask_to_bbox = ref_function(os.path.join(base, 'utils', 'geometry.py'), 'mask_to_bbox', {'np': np})
    pyco = types.SimpleNamespace(mask=types.SimpleNamespace(
        frPyObjects=lambda segs, h, w: list(segs),
        decode=lambda rles: np.stack([np_data.rle_decode(r) for r in rles], axis=2)))
    ns = {'np': np, 'PIL': PIL, 'pycocotools': pyco,
          'utils': types.SimpleNamespace(mask_to_bbox=mask_to_bbox)}
    fn = ref_function(os.path.join(base, 'datasets', 'coco.py'), '_annotations_to_example', ns)
    cat_ids = {c['id']: i for i, c in enumerate(sorted(COCO_CATEGORIES, key=lambda x: x['id']))}
    fx = {'annotations_json': np.array(json.dumps(anns)), 'height': H, 'width': W,
          'categories_json': np.array(json.dumps(COCO_CATEGORIES))}
    for tag, use_crowd in (('nocrowd', False), ('crowd', True)):
        self = types.SimpleNamespace(_use_crowd=use_crowd, _return_crowd=True, _return_area=True,
                                     cat_id_to_class_id=cat_ids)
        bboxes, labels, masks, crowds, areas = fn(self, anns, H, W)
        fx.update({tag + '_bboxes': bboxes, tag + '_labels': labels,
                   tag + '_masks': np.packbits(masks.astype(np.uint8), axis=-1),
                   tag + '_masks_shape': np.array(masks.shape), tag + '_crowds': crowds,
                   tag + '_areas': areas})
    return fx


def detectron_blobs(seed=5, n_layers=50):
    """Seeded stand-in for Detectron's `model_final.pkl['blobs']`: every blob the converter
    reads, with its real shape, plus the kinds it must ignore (momentum, fc1000, conv biases)."""
    import zlib
    blocks = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3)}[n_layers]
    shapes = {'conv1_w': (64, 3, 7, 7), 'conv1_b': (64,), 'res_conv1_bn_s': (64,),
              'res_conv1_bn_b': (64,), 'conv_rpn_w': (1024, 1024, 3, 3), 'conv_rpn_b': (1024,),
              'rpn_bbox_pred_w': (60, 1024, 1, 1), 'rpn_bbox_pred_b': (60,),
              'rpn_cls_logits_w': (15, 1024, 1, 1), 'rpn_cls_logits_b': (15,),
              'cls_score_w': (81, 2048), 'cls_score_b': (81,), 'bbox_pred_w': (324, 2048),
              'bbox_pred_b': (324,), 'conv5_mask_w': (2048, 256, 2, 2), 'conv5_mask_b': (256,),
              'mask_fcn_logits_w': (81, 256, 1, 1), 'mask_fcn_logits_b': (81,),
              'fc1000_w': (10, 2048), 'fc1000_b': (10,), 'conv1_w_momentum': (64, 3, 7, 7)}
    cin = 64
    for stage, n, mid, cout in zip((2, 3, 4, 5), blocks, (64, 128, 256, 512), (256, 512, 1024, 2048)):
        for i in range(n):
            ci = cin if i == 0 else cout
            pre = 'res%d_%d_' % (stage, i)
            for br, shp in (('branch2a', (mid, ci, 1, 1)), ('branch2b', (mid, mid, 3, 3)),
                            ('branch2c', (cout, mid, 1, 1))) + \
                    ((('branch1', (cout, ci, 1, 1)),) if i == 0 else ()):
                shapes[pre + br + '_w'] = shp
                shapes[pre + br + '_bn_s'] = (shp[0],)
                shapes[pre + br + '_bn_b'] = (shp[0],)
                shapes[pre + br + '_b'] = (shp[0],)
        cin = cout
    blobs = {}
    for k, shp in shapes.items():
        rng = np.random.RandomState((zlib.crc32(k.encode()) + seed) % (2 ** 31))
        blobs[k] = rng.standard_normal(shp).astype(np.float32)
    return blobs


def array_checksums(a):
    """(sum, sum of squares, position-weighted sum) in float64: any permutation, flip or slice
    error changes at least the third."""
    v = np.asarray(a, np.float64).ravel()
    w = ((np.arange(v.size, dtype=np.uint64) * np.uint64(2654435761)) % np.uint64(1000003)).astype(np.float64) / 1e6
    return np.array([v.sum(), (v * v).sum(), (v * w).sum()])


def detectron_fixture():
    import ast
    path = '/root/reference/examples/coco/convert_caffe2_to_chainer.py'
    tree = ast.parse(open(path).read())
    body, on = [], False
    for node in tree.body:
        tgt = node.targets[0].id if isinstance(node, ast.Assign) and isinstance(node.targets[0], ast.Name) else None
        if tgt == 'params_src':
            break
        if on:
            body.append(node)
        if tgt == 'model':
            on = True
    blobs = detectron_blobs()

    class Leaf(object):
        def __init__(self, shape):
            self.data = np.full(shape, np.nan, np.float32)

    class Node(object):
        pass
    from oracle import np_step
    shapes = dict(np_step.param_shapes(50))
    for k in ('rpn.loc_score.W', 'rpn.loc_score.b', 'head.cls_loc_score.W', 'head.cls_loc_score.b'):
        del shapes[k]
    shapes.update({'rpn.loc.W': (60, 1024, 1, 1), 'rpn.loc.b': (60,), 'rpn.score.W': (15, 1024, 1, 1),
                   'rpn.score.b': (15,), 'head.cls_loc.W': (324, 2048), 'head.cls_loc.b': (324,),
                   'head.score.W': (81, 2048), 'head.score.b': (81,)})
    model = Node()
    for name, shp in shapes.items():
        cur = model
        parts = name.split('.')
        for part in parts[:-1]:
            if not hasattr(cur, part):
                setattr(cur, part, Node())
            cur = getattr(cur, part)
        setattr(cur, parts[-1], Leaf(shp))
    exec(compile(ast.Module(body=body, type_ignores=[]), path, 'exec'),
         {'np': np, 'blobs': blobs, 'model': model})
    fx = {}
    for name in sorted(shapes):
        cur = model
        for part in name.split('.'):
            cur = getattr(cur, part)
        assert not np.isnan(cur.data).any(), name + ' was not filled by the reference converter'
        fx[name.replace('.', '/')] = array_checksums(cur.data)
    return fx


TRAIN_STEP_CFG = dict(n_layers=50, H=160, W=224, batch=2, n_gt=3, n_sample=32, input_seed=3,
                      param_seed=0, np_random_seed=11,
                      proposal_creator_params=dict(min_size=0, n_train_pre_nms=600,
                                                   n_train_post_nms=100))


def train_step_margins(np_step, cfg):
    """Smallest ReLU margin (np_step.RELU_MARGINS) of the fixture step under ``cfg``: (value, site)."""
    P = np_step.synthetic_params(cfg['n_layers'], seed=cfg['param_seed'])
    imgs, bboxes, labels, masks, scales = np_step.synthetic_inputs(
        cfg['input_seed'], cfg['batch'], cfg['H'], cfg['W'], n_gt=cfg['n_gt'], scale=1.0)
    np.random.seed(cfg['np_random_seed'])
    np_step.RELU_MARGINS = m = {}
    try:
        np_step.train_step(P, imgs, bboxes, labels, masks, scales, n_layers=cfg['n_layers'],
                           n_sample=cfg['n_sample'], backward=False,
                           proposal_creator_params=cfg['proposal_creator_params'])
    finally:
        np_step.RELU_MARGINS = None
    site = min(m, key=m.get)
    return m[site], site


def search_train_step_seeds(n=48):
    """`python oracle/gen_golden.py --search-train-step-seeds [n]`: how TRAIN_STEP_CFG's seeds were
    chosen.  The fixture is compared ENTRY BY ENTRY (1e-4 of a tensor's scale) with other fp32-class
    implementations, which is well posed only if no ReLU with a backward sits on a pre-activation
    within rounding of zero (one flipped decision moves a patch of every gradient below it by up to
    1e-3).  fp32 rounding of these layers is ~1e-7 .. 1e-6 of the site's scale; among ~3e6 units the
    typical smallest margin is of that order, so the seeds are searched for the LARGEST smallest
    margin.  Prints every candidate; edit TRAIN_STEP_CFG by hand and regenerate."""
    from oracle import np_step
    rows = []
    for k in range(n):
        cfg = dict(TRAIN_STEP_CFG, input_seed=1 + k, np_random_seed=11 + k)
        v, site = train_step_margins(np_step, cfg)
        rows.append((v, cfg['input_seed'], cfg['np_random_seed'], site))
        print('input_seed %3d np_random_seed %3d: smallest ReLU margin %.3e at %s'
              % (cfg['input_seed'], cfg['np_random_seed'], v, site), flush=True)
    rows.sort(reverse=True)
    print('best:', rows[:5])
    return rows


def train_step_fixture(np_step):
    c = TRAIN_STEP_CFG
    P = np_step.synthetic_params(c['n_layers'], seed=c['param_seed'])
    imgs, bboxes, labels, masks, scales = np_step.synthetic_inputs(
        c['input_seed'], c['batch'], c['H'], c['W'], n_gt=c['n_gt'], scale=1.0)
    np.random.seed(c['np_random_seed'])
    out = np_step.train_step(P, imgs, bboxes, labels, masks, scales, n_layers=c['n_layers'],
                             n_sample=c['n_sample'],
                             proposal_creator_params=c['proposal_creator_params'])
    fx = {'loss_names': np.array(sorted(out['losses'])),
          'loss_values': np.array([out['losses'][k] for k in sorted(out['losses'])], np.float64),
          'n_rois': np.array([len(r) for r in out['rois']], np.int32),
          'rois': np.concatenate(out['rois'], 0),
          'roi_order': np.concatenate(out['roi_order'], 0).astype(np.int32),
          'sample_rois': out['sample_rois'], 'sample_roi_indices': out['sample_roi_indices'],
          'gt_roi_labels': out['gt_roi_labels'], 'gt_roi_masks': out['gt_roi_masks'].astype(np.int8),
          'gt_rpn_labels': out['gt_rpn_labels'].astype(np.int8),
          'np_random_after': np.array(np.random.randint(0, 2 ** 31 - 1))}
    names = sorted(out['grads'])
    fx['grad_names'] = np.array(names)
    fx['grad_l2'] = np.array([np.sqrt(np.sum(out['grads'][k].astype(np.float64) ** 2)) for k in names])
    fx['grad_absmax'] = np.array([np.abs(out['grads'][k]).max() for k in names], np.float64)
    for k in ('rpn.loc_score.b', 'rpn.conv1.b', 'head.cls_loc_score.b', 'head.deconv6.b',
              'head.mask.b', 'extractor.res3.a.conv1.W'):
        fx['grad/' + k] = out['grads'][k].astype(np.float32)
    return fx


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == '--search-train-step-seeds':
        search_train_step_seeds(int(sys.argv[2]) if len(sys.argv) > 2 else 48)
    else:
        main()
