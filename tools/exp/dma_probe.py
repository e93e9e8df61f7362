import ctypes, os, torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libdma.so'))
lib.dma_probe.argtypes = [ctypes.c_void_p, ctypes.c_uint, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
src = torch.arange(1024, dtype=torch.float32, device='cuda') + 1
out = torch.zeros(1024, device='cuda')
for oob in (-1, 5):
    lib.dma_probe(src.data_ptr(), 4096, out.data_ptr(), oob, None)
    torch.cuda.synchronize()
    o = out.cpu()
    ok = torch.equal(o, src.cpu()) if oob < 0 else None
    print('oob lane', oob, 'linear ok' if ok else '', 'lane-5 slots of each wave:', [o[w * 256 + 20:w * 256 + 24].tolist() for w in range(4)],
          'mismatches elsewhere:', int(((o != src.cpu()).sum())))
