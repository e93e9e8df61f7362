"""Developer probe: package power, shader clock and temperature (rocm-smi samples) while one GEMM
shape runs in a loop on each arithmetic of the convolution kernels."""
import json, os, subprocess, sys, threading, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from chainer_mask_rcnn_amd import _lib
from chainer_mask_rcnn_amd.functions import conv as C
from chainer_mask_rcnn_amd.functions._layout import empty_nhwc

dev = torch.device('cuda:0')


def sample(stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(['rocm-smi', '--showpower', '--showclocks', '--showtemp', '--json'],
                               capture_output=True, text=True, timeout=10)
            d = json.loads(r.stdout)
            card = next(iter(d.values()))
            out.append({k: v for k, v in card.items()
                        if any(s in k.lower() for s in ('power', 'sclk', 'junction', 'edge'))})
        except Exception as e:        # noqa
            out.append({'error': str(e)})
        time.sleep(0.3)


def main():
    lib = _lib.load()
    N, Cc, H, W, K = 1024, 2048, 7, 7, 512
    x = torch.randn((N, H, W, Cc), device=dev).permute(0, 3, 1, 2)
    w = (torch.randn((K, 1, 1, Cc), device=dev) * 0.05).permute(0, 3, 1, 2)
    d = C.make_desc(x.shape, w.shape, 1, 0)
    y = empty_nhwc((d.N, d.K, d.P, d.Q), dev)
    flop = 2.0 * d.N * d.P * d.Q * K * Cc
    for kind, data in (('idle', None), ('fp32', 'random'), ('split_bf16x3', 'random'), ('split_bf16x3', 'zeros')):
        if kind != 'idle':
            C.set_gemm_arithmetic(kind)
            if data == 'zeros':
                x.zero_()
        stop, out = threading.Event(), []
        th = threading.Thread(target=sample, args=(stop, out))
        th.start()
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < 6.0:
            if kind == 'idle':
                time.sleep(0.2)
                continue
            for _ in range(200):
                _lib.call('mrcnn_conv2d_fwd', C.ctx_desc(d), _lib.ptr(x), _lib.ptr(w), None, None, None,
                          None, _lib.ptr(y), 0, _lib.ptr(C.split_ws(dev)), _lib.stream_ptr())
            torch.cuda.synchronize()
            n += 200
        el = time.perf_counter() - t0
        stop.set(); th.join()
        tf = flop * n / el / 1e12 if n else 0.
        print('== %s / %s operands: %.1f TFLOP/s over %.1f s; rocm-smi samples (last 6):' % (kind, data, tf, el))
        for s in out[-6:]:
            print('   ', s)
    C.set_gemm_arithmetic(C.DEFAULT_GEMM_ARITHMETIC)


if __name__ == '__main__':
    main()
