"""Does a 32-channel slice of a wide NHWC buffer stream at HBM rate?  (experiment aid)
affine_accumulate on channels [c0, c0+32) of buffers with different channel counts (= pixel pitch): a power-of-two pitch puts every
64-byte piece of the slice on the same few HBM channels."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fd-gan_amd")]
import torch
from fdgan_hip import engine as E
dev = "cuda:0"
N, H, W = 16, int(sys.argv[1]) if len(sys.argv) > 1 else 128, int(sys.argv[1]) if len(sys.argv) > 1 else 128
C = int(sys.argv[2]) if len(sys.argv) > 2 else 32
P = N * H * W
coef = torch.randn(2, 2048, device=dev) * 1e-3
for ctot in (C, 256, 256 + 32, 256 + 64, 512, 512 + 32, 512 + 64, 512 + 128, 1024, 1024 + 32, 1024 + 64, 1024 + 128, 1024 + 256):
    if ctot < C: continue
    x = torch.randn(N, H, W, ctot, device=dev).half()
    g = torch.randn(N, H, W, ctot, device=dev).bfloat16()
    c0 = min(ctot - C, 96)
    xv, gv = E.View(x, c0, C), E.View(g, c0, C)
    for _ in range(3): E.affine_accumulate(xv.fd, coef[0, :C], coef[1, :C], gv.fd)
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(20): E.affine_accumulate(xv.fd, coef[0, :C], coef[1, :C], gv.fd)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print("pitch %5d B  slice %3d ch @%dx%d: %7.1f us  %5.2f TB/s" % (ctot * 2, C, H, W, us, P * C * 6 / us / 1e6))
    del x, g
