"""Pin the C oracle (oracle/roi_align_ref.c) against vectors produced by the
REFERENCE's own forward_cpu / backward_cpu (tests/golden, oracle/gen_golden.py)."""
import glob
import os

import numpy as np
import pytest

import oracle

# north_star tolerance for fp32 ROIAlign: 1e-4 relative
RTOL, ATOL = 1e-4, 1e-5


def _cases(golden_dir):
    return sorted(glob.glob(os.path.join(golden_dir, 'roi_align_*.npz')))


def test_golden_present(golden_dir):
    assert len(_cases(golden_dir)) == 7


@pytest.mark.parametrize('name', ['roi_align_testgeom_sr0', 'roi_align_testgeom_sr1',
                                  'roi_align_testgeom_sr2', 'roi_align_toy0',
                                  'roi_align_toy1', 'roi_align_toy2', 'roi_align_c4like'])
def test_oracle_matches_reference(golden_dir, name):
    d = np.load(os.path.join(golden_dir, name + '.npz'))
    rois = d['rois_yx'][:, [0, 2, 1, 4, 3]] if 'rois_yx' in d else d['rois']
    args = (int(d['outh']), int(d['outw']), float(d['spatial_scale']), int(d['sampling_ratio']))
    y = oracle.roi_align_fwd(d['x'], rois, *args)
    np.testing.assert_allclose(y, d['y'], rtol=RTOL, atol=ATOL)
    gx = oracle.roi_align_bwd(d['gy'], rois, d['x'].shape, args[2], args[3])
    np.testing.assert_allclose(gx, d['gx'], rtol=RTOL, atol=ATOL * 10)


def test_gradient_mass_conserved(golden_dir):
    # sum(gx) == sum(gy) when no sample is skipped
    d = np.load(os.path.join(golden_dir, 'roi_align_c4like.npz'))
    rois = d['rois_yx'][:, [0, 2, 1, 4, 3]]
    gx = oracle.roi_align_bwd(d['gy'], rois, d['x'].shape, 1 / 16., 0)
    assert abs(gx.sum(dtype=np.float64) - d['gy'].sum(dtype=np.float64)) < 1e-2


def test_out_of_range_samples_are_skipped_not_hung():
    # GPU semantics (roi_align_2d.py:228-231,282): skipped samples, full count divisor.
    x = np.ones((1, 1, 4, 4), np.float32)
    rois = np.array([[0, 0, 0, 200, 200]], np.float32)
    y = oracle.roi_align_fwd(x, rois, 2, 2, 1.0, 2)
    assert np.isfinite(y).all()
    assert y[0, 0, 0, 0] < 1.0  # part of the samples fell outside and were dropped


def test_affine_golden(golden_dir):
    from oracle import np_ref
    d = np.load(os.path.join(golden_dir, 'affine_channel_2d.npz'))
    y = np_ref.affine_channel_2d_fwd(d['x'], d['W'], d['b'])
    np.testing.assert_allclose(y, d['y'], rtol=1e-6, atol=1e-7)
    gx, gW, gb = np_ref.affine_channel_2d_bwd(d['x'], d['W'], d['gy'])
    np.testing.assert_allclose(gx, d['gx'], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(gW, d['gW'].ravel(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(gb, d['gb'].ravel(), rtol=1e-5, atol=1e-5)
