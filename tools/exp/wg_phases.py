"""Developer experiment: are the workgroups co-resident on a CU in lockstep?  Needs the
-DMRCNN_GEMM_CLOCKPROBE library.  Prints, for a few CUs, the (start, end) of every workgroup
of ONE launch of the res5 conv3 forward (K=512, N=2048: 16 K slices, wide epilogue)."""
import sys, os, ctypes
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from chainer_mask_rcnn_amd import _lib
from chainer_mask_rcnn_amd.functions.conv import make_desc, ctx_desc, split_ws
from chainer_mask_rcnn_amd.functions._layout import empty_nhwc
dev = torch.device('cuda:0')
NSLOT = 16384
lib = _lib.load()
raw = ctypes.CDLL(_lib.LIB_PATH)
C, K = int(os.environ.get('CIN', 512)), int(os.environ.get('COUT', 2048))
N, H, W = 1024, 7, 7
x = torch.randn((N, H, W, C), device=dev).permute(0, 3, 1, 2)
w = (torch.randn((K, 1, 1, C), device=dev) * 0.05).permute(0, 3, 1, 2)
d = make_desc(x.shape, w.shape, 1, 0)
y = empty_nhwc((d.N, d.K, d.P, d.Q), dev)
sw, sp = _lib.ptr(split_ws(dev)), _lib.stream_ptr()
fn = lambda: _lib.call('mrcnn_conv2d_fwd', ctx_desc(d), _lib.ptr(x), _lib.ptr(w), None, None, None,
                       None, _lib.ptr(y), 0, sw, sp)
buf = (ctypes.c_ulonglong * (NSLOT * 5))()
for _ in range(3):
    fn()
torch.cuda.synchronize()
raw.mrcnn_gemm_probe_read(buf, NSLOT * 5)
fn()
torch.cuda.synchronize()
raw.mrcnn_gemm_probe_read(buf, NSLOT * 5)
buf2 = (ctypes.c_ulonglong * (NSLOT * 8))()
raw.mrcnn_gemm_probe2_read(buf2, NSLOT * 8)
t = np.frombuffer(buf, dtype=np.uint64).reshape(NSLOT, 5).astype(np.int64)
t2 = np.frombuffer(buf2, dtype=np.uint64).reshape(NSLOT, 8).astype(np.int64)
idx = np.flatnonzero(t[:, 2] > 0)
t = t[idx]
t2 = t2[idx]
pro, loop, epi = (t2[:, 0] - t[:, 1]) / 100.0, (t2[:, 1] - t2[:, 0]) / 100.0, (t[:, 3] - t2[:, 1]) / 100.0
main = idx < int(os.environ.get('MAIN_TILES', 6272))
st = np.diff(t2[main][:, 1:7], axis=1) / 100.0
print('epilogue stages p50 us: loop end->barrier %.2f, ->LDS(i=0) %.2f, ->stores(i=0) %.2f, ->LDS(i=1) %.2f, ->stores(i=1) %.2f' % tuple(np.median(st, axis=0)))
print('whole tiles: setup %.2f us (p50), first slice + K loop %.1f us, epilogue %.2f us; p90 %.2f / %.1f / %.2f'
      % (np.median(pro[main]), np.median(loop[main]), np.median(epi[main]),
         np.percentile(pro[main], 90), np.percentile(loop[main], 90), np.percentile(epi[main], 90)))
r0 = t[:, 1].min()
start, end = (t[:, 1] - r0) / 100.0, (t[:, 3] - r0) / 100.0      # us
xcc, hw = t[:, 4] >> 32, t[:, 4] & 0xffffffff
cu = (xcc << 8) | ((hw >> 8) & 0xff)
print('workgroups %d, kernel span %.1f us, distinct CU keys %d' % (len(t), end.max(), len(np.unique(cu))))
print('hw_id fields of the first 8 workgroups: ' + ' '.join('%d:%x/%08x' % (i, a, b) for i, a, b in zip(idx[:8], xcc[:8], hw[:8])))
for key in np.unique(cu)[:3]:
    m = np.flatnonzero(cu == key)
    order = m[np.argsort(start[m])]
    print('CU %04x: %d workgroups' % (key, len(m)))
    print('   ' + ' '.join('[%d %.0f-%.0f]' % (idx[i], start[i], end[i]) for i in order))
# lockstep metric: for each workgroup, how many others on the same CU end within 3 us of it
near = []
for key in np.unique(cu):
    m = np.flatnonzero(cu == key)
    e = np.sort(end[m])
    for v in e:
        near.append(np.sum(np.abs(e - v) < 3.0) - 1)
near = np.asarray(near)
print('workgroups whose end coincides (<3 us) with k others on the same CU: ' +
      ', '.join('k=%d: %.0f%%' % (k, 100.0 * np.mean(near == k)) for k in range(4)))
print('median workgroup duration %.1f us' % np.median(end - start))
