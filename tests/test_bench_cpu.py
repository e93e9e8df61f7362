"""bench.py pieces that run without a GPU: the power / clock sampler degrades to None fields, the
multi-GPU default drops the auxiliary legs, and a run without a ROCm device fails loudly."""
import os
import subprocess
import sys

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_smi_sampler_without_a_device_reports_none():
    s = bench.SmiSampler(0, period=0.001)
    with s:
        pass
    out = s.summary()
    assert set(out) == {'power_w', 'power_w_max', 'sclk_mhz', 'sclk_mhz_min', 'samples'}
    if s.lib is None or out['samples'] == 0 or out['power_w'] is None:
        assert out['power_w'] is None and out['power_w_max'] is None


def test_bench_refuses_to_run_without_a_rocm_device():
    import torch
    if torch.cuda.is_available():
        return                       # (a GPU box: nothing to refuse, and a full bench run is not a CPU test)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '1', '--warmup', '0',
                        '--no-cpu-baseline'], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and r.stdout.strip() == ''          # no JSON line from a run that measured nothing
    assert 'ROCm device' in r.stderr or 'HIP' in r.stderr or 'hip' in r.stderr


def test_smi_sampler_can_be_disabled_and_survives_a_library_without_the_symbols(monkeypatch):
    s = bench.SmiSampler(0, enabled=False)
    with s:
        pass
    assert s.lib is None and s.summary()['samples'] == 0

    class NoSymbols(object):          # a librocm_smi64 that lacks the two query functions
        def __getattr__(self, name):
            raise AttributeError(name)
    monkeypatch.setattr(bench.ctypes, 'CDLL', lambda name: NoSymbols())
    s = bench.SmiSampler(0)
    with s:
        pass
    assert s.lib is None and s.summary()['power_w'] is None


def test_multi_gpu_request_without_devices_is_refused():
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        return
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1'],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and 'one process per GPU' in (r.stderr + r.stdout)
