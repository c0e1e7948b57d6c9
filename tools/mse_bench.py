"""Times fdgan_mse_nhwc_fwd / _bwd on Vgg16's four tap shapes at B = 16 @ 256^2 (dense views: the flat kernel) and on the same
shapes as channel slices of a buffer 8 channels wider (the generic kernel).  Usage: python tools/mse_bench.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fd-gan_amd"))
from fdgan_hip import engine as E
from fdgan_hip import lib as L

dev = torch.device("cuda:0")
lib = L.load()
part = torch.empty(1 << 16, dtype=torch.float32, device=dev)
up = torch.ones((), dtype=torch.float32, device=dev)


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        fn()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) * 1e3 / reps


for (n, h, w, c) in ((16, 256, 256, 64), (16, 128, 128, 128), (16, 64, 64, 256), (16, 32, 32, 512)):
    for pad in (0, 8):
        a = (torch.randn(n, h, w, c + pad, device=dev) * 0.5).half()
        b = (torch.randn(n, h, w, c + pad, device=dev) * 0.5).half()
        g = torch.zeros(n, h, w, c + pad, device=dev, dtype=torch.bfloat16)
        av, bv, gv = E.View(a, 0, c), E.View(b, 0, c), E.View(g, 0, c)
        np_ = C.c_int64(0)
        fwd = lambda: L.check(lib.fdgan_mse_nhwc_fwd(C.byref(av.fd), C.byref(bv.fd), 1.0 / (n * h * w * c), part.data_ptr(), part.numel(),
                                                    C.byref(np_), E.stream_ptr()))
        bwd = lambda: L.check(lib.fdgan_mse_nhwc_bwd(C.byref(av.fd), C.byref(bv.fd), up.data_ptr(), 2.0 / (n * h * w * c), 1, C.byref(gv.fd),
                                                    E.stream_ptr()))
        tf, tb = timed(fwd), timed(bwd)
        val = float(part[:np_.value].double().sum())
        ref = float(((a[..., :c].float() - b[..., :c].float()) ** 2).mean())
        gref = (2.0 / (n * h * w * c)) * (a[..., :c].float() - b[..., :c].float()) * (a[..., :c] > 0)
        gerr = float((g[..., :c].float() - gref).abs().max() / gref.abs().max())
        mb = n * h * w * c * 2 / 1e6
        print("%2dx%3dx%3dx%3d %s  fwd %6.1f us (%4.2f TB/s)  bwd %6.1f us (%4.2f TB/s)  value %.6f vs torch %.6f, grad rel err %.1e"
              % (n, h, w, c, "generic" if pad else "flat   ", tf, 2 * mb / tf, tb, 3 * mb / tb, val, ref, gerr))
