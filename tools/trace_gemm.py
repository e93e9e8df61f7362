"""Developer tool: per-phase cycle breakdown of the GEMM K loop (needs a library built with
-DMRCNN_GEMM_TRACE)."""
import sys, os, ctypes
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chainer_mask_rcnn_amd import _lib
from chainer_mask_rcnn_amd.functions.conv import make_desc, ctx_desc
from chainer_mask_rcnn_amd.functions._layout import empty_nhwc
dev = torch.device('cuda:0')
lib = _lib.load()
N, C, H, W, K, k, s, p = 1024, 512, 7, 7, 512, 3, 1, 1
x = torch.randn((N, H, W, C), device=dev).permute(0, 3, 1, 2)
w = (torch.randn((K, k, k, C), device=dev) * 0.05).permute(0, 3, 1, 2)
d = make_desc(x.shape, w.shape, s, p)
y = empty_nhwc((d.N, d.K, d.P, d.Q), dev)
for _ in range(3):
    _lib.call('mrcnn_conv2d_fwd', ctx_desc(d), _lib.ptr(x), _lib.ptr(w), None, None, None, None,
              _lib.ptr(y), 0, None, _lib.stream_ptr())
torch.cuda.synchronize()
n = 64 * 4 * 64 * 5
buf = (ctypes.c_ulonglong * n)()
raw = ctypes.CDLL(_lib.LIB_PATH)
raw.mrcnn_gemm_trace_read(buf, n)
t = np.frombuffer(buf, dtype=np.uint64).reshape(64, 4, 64, 5).astype(np.int64)
names = ['load issue', 'compute (64 MFMA)', 'vmcnt wait', 'LDS store + barrier']
dt = np.diff(t, axis=3)                       # (blocks, waves, slices, 4)
nxt = t[:, :, 1:, 0] - t[:, :, :-1, 4]        # barrier exit -> next loop top
print('cycles per slice per wave (s_memtime ticks = shader cycles), mean over 64 blocks x 4 waves x 64 slices')
for i, nm in enumerate(names):
    print('%-22s mean %8.1f  p50 %8.1f  p90 %8.1f' % (nm, dt[..., i].mean(), np.median(dt[..., i]), np.percentile(dt[..., i], 90)))
tot = t[:, :, 1:, 0] - t[:, :, :-1, 0]
print('%-22s mean %8.1f  p50 %8.1f' % ('whole slice', tot.mean(), np.median(tot)))
print('MFMA floor per slice per wave: 4096 cycles; two waves share a SIMD -> 8192 per pair')
