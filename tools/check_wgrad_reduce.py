"""Developer check: the in-kernel slab reduction of the wgrad kernel (last-arriving workgroup
sums the slabs after an agent-scope release/acquire hand-off) must be bit-identical, launch
after launch and under load, to a reference sum of the same partials."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chainer_mask_rcnn_amd import _lib
from chainer_mask_rcnn_amd.functions.conv import make_desc, ctx_desc

dev = torch.device('cuda:0')
lib = _lib.load()
torch.manual_seed(0)
bad = 0
for (N, C, H, W, K, k) in [(1024, 512, 7, 7, 512, 3), (1024, 512, 7, 7, 2048, 1), (2, 1024, 51, 84, 1024, 3),
                           (2, 256, 51, 84, 256, 3), (2, 128, 101, 167, 512, 1)]:
    w = torch.empty((K, k, k, C), device=dev).permute(0, 3, 1, 2)
    sets = []
    for i in range(3):          # three different problems share the workspace in turn, so a
        x = torch.randn((N, H, W, C), device=dev).permute(0, 3, 1, 2)     # stale slab would show
        gy = torch.randn((N, H, W, K), device=dev).permute(0, 3, 1, 2) * (i + 1)
        sets.append((x, gy))
    d = make_desc(sets[0][0].shape, w.shape, 1, k // 2)
    ws = _lib.workspace(lib.mrcnn_conv2d_wgrad_workspace_bytes(ctx_desc(d)), dev, 'wgrad')
    refs = []
    for x, gy in sets:          # reference: no workspace -> no split, one workgroup per tile
        gw = torch.empty((K, k, k, C), device=dev)
        _lib.call('mrcnn_conv2d_wgrad', ctx_desc(d), _lib.ptr(x), _lib.ptr(gy), _lib.ptr(gw),
                  None, _lib.stream_ptr())
        refs.append(gw)
    side = torch.cuda.Stream()
    outs = []
    for it in range(60):
        x, gy = sets[it % 3]
        gw = torch.full((K, k, k, C), float('nan'), device=dev)
        if it % 2:      # uneven load from another stream while the kernel runs
            with torch.cuda.stream(side):
                junk = torch.randn(4096, 4096, device=dev) @ torch.randn(4096, 4096, device=dev)
        _lib.call('mrcnn_conv2d_wgrad', ctx_desc(d), _lib.ptr(x), _lib.ptr(gy), _lib.ptr(gw),
                  _lib.ptr(ws), _lib.stream_ptr())
        outs.append(gw)
    torch.cuda.synchronize()
    worst, same = 0., True
    for it, o in enumerate(outs):
        r = refs[it % 3]
        worst = max(worst, ((o - r).abs().max() / r.abs().max()).item())
        same &= torch.equal(o, outs[it % 3])
    print((N, C, H, W, K, k), 'max rel err vs unsplit: %.2e' % worst, 'repeatable:', same)
    bad += (not same) or not (worst < 1e-4)
print('FAIL' if bad else 'OK')
