"""The C-ABI library builds for gfx950, loads, and exports every symbol that
include/mrcnn_hip.h declares (no compute calls: there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'mrcnn_hip.h')


@pytest.fixture(scope='module')
def lib():
    import __graft_entry__ as g
    g.build()
    from chainer_mask_rcnn_amd import _lib
    return _lib.load()


def _declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(mrcnn_[a-z0-9_]+)\s*\(', text)))


def test_header_declares_the_path():
    syms = _declared_symbols()
    for must in ['mrcnn_roi_align_fwd', 'mrcnn_roi_align_bwd', 'mrcnn_nms_sorted',
                 'mrcnn_conv2d_fwd', 'mrcnn_conv2d_dgrad', 'mrcnn_conv2d_wgrad',
                 'mrcnn_deconv2x2s2_fwd', 'mrcnn_sgd_momentum_wd']:
        assert must in syms


def test_every_declared_symbol_is_exported(lib):
    for s in _declared_symbols():
        assert hasattr(lib, s), 'libmrcnn_hip.so does not export %s' % s


def test_binding_table_covers_header(lib):
    from chainer_mask_rcnn_amd import _lib
    assert sorted(_lib.SIGNATURES) == _declared_symbols()


def test_abi_version_and_error_string(lib):
    assert lib.mrcnn_abi_version() == 1
    assert isinstance(lib.mrcnn_last_error(), bytes)


def test_argument_validation_without_device(lib):
    # pure host-side checks return an error code and a message, never crash
    rc = lib.mrcnn_roi_align_fwd(None, None, None, 1, 4, 4, 4, 1, 2, 2,
                                 ctypes.c_float(1.0), -1, None)
    assert rc != 0 and b'sampling_ratio' in lib.mrcnn_last_error()
    from chainer_mask_rcnn_amd._lib import ConvDesc
    d = ConvDesc(1, 8, 8, 6, 8, 3, 3, 1, 1, 8, 8)   # C=6 not a multiple of 4
    rc = lib.mrcnn_conv2d_fwd(ctypes.byref(d), None, None, None, None, None, None, None, 0, None,
                              None)
    assert rc != 0 and b'multiples of 4' in lib.mrcnn_last_error()
    buf = ctypes.create_string_buffer(64)
    # the ROIAlign backward refuses a workspace smaller than its own size query
    need = lib.mrcnn_roi_align_bwd_workspace_bytes(1, 8, 8, 4, 7, 7, 1)
    assert need > 0
    ws = ctypes.create_string_buffer(64)
    rc = lib.mrcnn_roi_align_bwd_ws(ctypes.cast(buf, ctypes.c_void_p), ctypes.cast(buf, ctypes.c_void_p),
                                    ctypes.cast(buf, ctypes.c_void_p), 1, 8, 8, 4, 4, 7, 7, 1,
                                    ctypes.c_float(0.0625), 0, ctypes.cast(ws, ctypes.c_void_p), need - 1, None)
    assert rc != 0 and b'workspace smaller' in lib.mrcnn_last_error()


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'chainer_mask_rcnn_amd')
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith('.py'):
                src = open(os.path.join(dp, fn)).read()
                assert 'import oracle' not in src and 'from oracle' not in src, fn


def test_ops_fail_loudly_on_cpu_tensors():
    import torch
    from chainer_mask_rcnn_amd import functions
    from chainer_mask_rcnn_amd._lib import MrcnnHipError
    x = torch.zeros(1, 4, 4, 4)
    rois = torch.tensor([[0., 0., 0., 2., 2.]])
    with pytest.raises(MrcnnHipError):
        functions.roi_align_2d(x, rois, 2, 2, 1.0)


def test_roi_align_argument_errors():
    # error conventions of roi_align_2d.py:30-43, :555-556
    import torch
    from chainer_mask_rcnn_amd import functions
    with pytest.raises(TypeError):
        functions.ROIAlign2D(2.0, 2, 1.0)
    with pytest.raises(TypeError):
        functions.ROIAlign2D(2, 2, 1.0, sampling_ratio=-1)
    with pytest.raises(TypeError):
        functions.ROIAlign2D(2, 2, 'a')
    assert functions.ROIAlign2D(2, 2, 1).spatial_scale == 1.0
    with pytest.raises(ValueError):
        functions.roi_align_2d(torch.zeros(1, 1, 2, 2), torch.zeros(1, 5), 2, 2, 1.0, axes='zz')
    with pytest.raises(TypeError):
        functions.ROIAlign2D(2, 2, 1.0)(torch.zeros(1, 1, 2, 2), torch.zeros(1, 4))


def test_bench_refuses_more_ranks_than_devices():
    """`python bench.py --gpus N` self-launches one rank per GPU (SCALE runs); with fewer than
    N devices visible it must fail loudly BEFORE any rendezvous, not hang or oversubscribe."""
    import subprocess
    import sys
    env = dict(os.environ)
    env.pop('WORLD_SIZE', None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '64'],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0
    assert 'ROCm device(s) visible' in r.stderr and r.stdout.strip() == ''


def test_kernel_selection_switches_without_device(lib):
    """mrcnn_set_tuning: the arithmetic switch of the GEMM kernels and the host mirror's wrapper."""
    from chainer_mask_rcnn_amd import _lib
    from chainer_mask_rcnn_amd.functions import conv
    assert lib.mrcnn_set_tuning(b'split_bf16', 0) == 0
    assert lib.mrcnn_set_tuning(b'split_bf16', 3) == 0
    rc = lib.mrcnn_set_tuning(b'no_such_switch', 1)
    assert rc != 0 and b'unknown option' in lib.mrcnn_last_error()
    with pytest.raises(_lib.MrcnnHipError):
        _lib.set_tuning('no_such_switch', 1)
    with pytest.raises(ValueError):
        conv.set_gemm_arithmetic('bf16')
    assert conv.DEFAULT_GEMM_ARITHMETIC == 'split_bf16x3'
    conv.set_gemm_arithmetic('fp32')
    assert conv.GEMM_ARITHMETIC == 'fp32'
    conv.set_gemm_arithmetic('split_bf16x3')
    assert conv.GEMM_ARITHMETIC == 'split_bf16x3'


def test_roi_spatial_order_is_a_permutation_grouped_by_image_and_band():
    """Host helper behind roi_align_2d(order=): sorted by (image, 6-row band of the centre, x centre)."""
    import numpy as np
    from chainer_mask_rcnn_amd import functions
    rng = np.random.RandomState(0)
    R = 200
    y1 = rng.uniform(0, 700, R); x1 = rng.uniform(0, 1200, R)
    rois = np.stack([y1, x1, y1 + rng.uniform(8, 300, R), x1 + rng.uniform(8, 300, R)], 1).astype(np.float32)
    idx = rng.randint(0, 2, R)
    o = functions.roi_spatial_order(rois, idx, 1 / 16.)
    assert o.dtype == np.int32 and sorted(o.tolist()) == list(range(R))
    yc = (rois[:, 0] + rois[:, 2]) * (0.5 / 16.)
    xc = (rois[:, 1] + rois[:, 3]) * (0.5 / 16.)
    keys = [(int(idx[i]), int(np.floor(yc[i] / 6.0)), float(xc[i])) for i in o]
    assert keys == sorted(keys)
    assert functions.roi_spatial_order(np.zeros((0, 4), np.float32), np.zeros((0,), np.int32), 1 / 16.).shape == (0,)


def test_mrcnn_tune_failure_is_sticky():
    """MRCNN_TUNE: every entry is parsed before any is applied, and a knob the library rejects makes
    every later load() raise the same error (the process-wide library state is half-tuned) instead
    of silently handing out the library on the second call."""
    import subprocess
    import sys
    code = (
        "import os, sys\n"
        "sys.path.insert(0, %r)\n"
        "from chainer_mask_rcnn_amd import _lib\n"
        "msgs = []\n"
        "for _ in range(2):\n"
        "    try:\n"
        "        _lib.load(); msgs.append('loaded')\n"
        "    except _lib.MrcnnHipError as e:\n"
        "        msgs.append(str(e))\n"
        "assert msgs[0] == msgs[1] and 'MRCNN_TUNE' in msgs[0], msgs\n"
        "print('ok')\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for tune in ('fused_tail=512,no_such_knob=1', 'fused_tail', '=3'):
        env = dict(os.environ, MRCNN_TUNE=tune)
        out = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True)
        assert out.returncode == 0 and out.stdout.strip() == 'ok', (tune, out.stdout, out.stderr)
    env = dict(os.environ, MRCNN_TUNE='fused_tail=512')
    ok = subprocess.run([sys.executable, '-c', code.replace("'MRCNN_TUNE' in msgs[0]", "msgs[0] == 'loaded'")],
                        env=env, capture_output=True, text=True)
    assert ok.returncode == 0, (ok.stdout, ok.stderr)
