"""Developer experiment: per-phase cycles of the single-stage K loop (library built with
-DMRCNN_GEMM_TRACE): barrier + wait for the slice's loads | LDS store + barrier | load issue |
64 MFMAs.  usage: MRCNN_HIP_LIB=.../libtrace.so python tools/exp/trace_singlebuf.py [fwd|wgrad] [C] [K]"""
import sys, os, ctypes
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from chainer_mask_rcnn_amd import _lib
from chainer_mask_rcnn_amd.functions.conv import make_desc, ctx_desc, split_ws
from chainer_mask_rcnn_amd.functions._layout import empty_nhwc
dev = torch.device('cuda:0')
lib = _lib.load()
mode = sys.argv[1] if len(sys.argv) > 1 else 'fwd'
C = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
K = int(sys.argv[3]) if len(sys.argv) > 3 else 512
N, H, W = 1024, 7, 7
x = torch.randn((N, H, W, C), device=dev).permute(0, 3, 1, 2)
w = (torch.randn((K, 1, 1, C), device=dev) * 0.05).permute(0, 3, 1, 2)
d = make_desc(x.shape, w.shape, 1, 0)
y = empty_nhwc((d.N, d.K, d.P, d.Q), dev)
gy = torch.randn((d.N, d.P, d.Q, d.K), device=dev).permute(0, 3, 1, 2)
gw = torch.empty_like(w)
ws = _lib.workspace(lib.mrcnn_conv2d_wgrad_workspace_bytes(ctx_desc(d)), dev, 'wgrad')
sp, sw = _lib.stream_ptr(), _lib.ptr(split_ws(dev))
if mode == 'fwd':
    fn = lambda: _lib.call('mrcnn_conv2d_fwd', ctx_desc(d), _lib.ptr(x), _lib.ptr(w), None, None, None,
                           None, _lib.ptr(y), 0, sw, sp)
else:
    fn = lambda: _lib.call('mrcnn_conv2d_wgrad', ctx_desc(d), _lib.ptr(x), _lib.ptr(gy), _lib.ptr(gw),
                           _lib.ptr(ws), sp)
for _ in range(3):
    fn()
torch.cuda.synchronize()
n = 64 * 4 * 64 * 5
buf = (ctypes.c_ulonglong * n)()
raw = ctypes.CDLL(_lib.LIB_PATH)
raw.mrcnn_gemm_trace_read(buf, n)
t = np.frombuffer(buf, dtype=np.uint64).reshape(64, 4, 64, 5).astype(np.int64)
ok = (t[..., 4] > 0) & (t[..., 0] > 0)
names = ['barrier + wait for loads', 'LDS store + barrier', 'load issue', 'compute (64 MFMA)']
dt = np.diff(t, axis=3)
print('%s C=%d K=%d: cycles per slice per wave, over %d stamped slices' % (mode, C, K, int(ok.sum())))
for i, nm in enumerate(names):
    v = dt[..., i][ok]
    print('%-26s mean %8.1f  p50 %8.1f  p90 %8.1f' % (nm, v.mean(), np.median(v), np.percentile(v, 90)))
tot = (t[:, :, 1:, 0] - t[:, :, :-1, 0])[ok[:, :, 1:] & ok[:, :, :-1]]
print('%-26s mean %8.1f  p50 %8.1f   (MFMA floor with three waves per SIMD: 12288)' % ('whole slice', tot.mean(), np.median(tot)))
