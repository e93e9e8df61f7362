"""Developer experiment: sustained shader clock and per-workgroup time of the GEMM kernels in
production conditions (library built with -DMRCNN_GEMM_CLOCKPROBE: every workgroup stamps
s_memtime / s_memrealtime at entry and exit).  usage:
    MRCNN_HIP_LIB=chainer_mask_rcnn_amd/csrc/variants/libclockprobe.so python tools/exp/clock_probe.py
"""
import sys, os, ctypes
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from chainer_mask_rcnn_amd import _lib
from chainer_mask_rcnn_amd.functions.conv import make_desc, ctx_desc, split_ws
from chainer_mask_rcnn_amd.functions._layout import empty_nhwc

dev = torch.device('cuda:0')
SHAPES = [
    ('res5 1x1 512->2048', 1024, 512, 7, 7, 2048, 1, 1, 0),
    ('res5 1x1 2048->512', 1024, 2048, 7, 7, 512, 1, 1, 0),
    ('res5 3x3 512', 1024, 512, 7, 7, 512, 3, 1, 1),
    ('res4 3x3 256', 2, 256, 51, 84, 256, 3, 1, 1),
    ('res4 1x1 256->1024', 2, 256, 51, 84, 1024, 1, 1, 0),
]
NSLOT = 16384


def read_probe(raw):
    buf = (ctypes.c_ulonglong * (NSLOT * 5))()
    raw.mrcnn_gemm_probe_read(buf, NSLOT * 5)
    return np.frombuffer(buf, dtype=np.uint64).reshape(NSLOT, 5).astype(np.int64)


def main():
    lib = _lib.load()
    raw = ctypes.CDLL(_lib.LIB_PATH)
    iters = int(os.environ.get('ITERS', 30))
    print('%-22s %-9s %7s %6s %8s %9s %9s %8s' % ('shape', 'pass', 'ms', 'TF/s', 'clk GHz', 'wg us p50',
                                                   'wg cyc p50', 'kernel us'))
    for name, N, C, H, W, K, k, s, p in SHAPES:
        x = torch.randn((N, H, W, C), device=dev).permute(0, 3, 1, 2)
        w = (torch.randn((K, k, k, C), device=dev) * 0.05).permute(0, 3, 1, 2)
        d = make_desc(x.shape, w.shape, s, p)
        y = empty_nhwc((d.N, d.K, d.P, d.Q), dev)
        gy = torch.randn((d.N, d.P, d.Q, d.K), device=dev).permute(0, 3, 1, 2)
        gx = empty_nhwc((N, C, H, W), dev)
        gw = torch.empty_like(w)
        ws = _lib.workspace(lib.mrcnn_conv2d_wgrad_workspace_bytes(ctx_desc(d)), dev, 'wgrad')
        sp = _lib.stream_ptr()
        sw = _lib.ptr(split_ws(dev))
        flop = 2.0 * d.N * d.P * d.Q * K * C * k * k
        wT = torch.empty((C * k * k * K,), device=dev)
        _lib.call('mrcnn_filter_flip_transpose', _lib.ptr(w), _lib.ptr(wT), K, k, k, C, None, sp)
        passes = [
            ('fwd', lambda: _lib.call('mrcnn_conv2d_fwd', ctx_desc(d), _lib.ptr(x), _lib.ptr(w), None,
                                      None, None, None, _lib.ptr(y), 0, sw, sp)),
            ('wgrad', lambda: _lib.call('mrcnn_conv2d_wgrad', ctx_desc(d), _lib.ptr(x), _lib.ptr(gy),
                                        _lib.ptr(gw), _lib.ptr(ws), sp)),
            ('dgrad_wt', lambda: _lib.call('mrcnn_conv2d_dgrad_wt', ctx_desc(d), _lib.ptr(gy), _lib.ptr(wT),
                                           _lib.ptr(gx), 0, None, None, None, None, None, None, sw, sp)),
            ('dgrad', lambda: _lib.call('mrcnn_conv2d_dgrad', ctx_desc(d), _lib.ptr(gy), _lib.ptr(w),
                                        _lib.ptr(gx), 0, sp)),
        ]
        for pname, fn in passes:
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            read_probe(raw)                     # (clears the stamps)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(iters):
                fn()
            b.record()
            torch.cuda.synchronize()
            ms = a.elapsed_time(b) / iters
            t = read_probe(raw)
            t = t[t[:, 2] > 0]
            dc, dr = t[:, 2] - t[:, 0], t[:, 3] - t[:, 1]
            ok = dr > 50                        # >= 0.5 us of reference clock for a usable ratio
            clk = np.median(dc[ok] / dr[ok]) * 0.1 if ok.any() else float('nan')   # 100 MHz ref
            span = (t[:, 3].max() - t[:, 1].min()) / 100.0
            print('%-22s %-9s %7.3f %6.1f %8.3f %9.1f %9.0f %8.1f   wgs %d' % (
                name, pname, ms, flop / ms / 1e9, clk, np.median(dr) / 100.0, np.median(dc), span, len(t)))
            # zero the probe area for the next pass
    print('(clock = median over workgroups of d(s_memtime)/d(s_memrealtime) x 100 MHz; kernel us = '
          'first entry to last exit of the LAST launch of the loop)')


if __name__ == '__main__':
    main()
