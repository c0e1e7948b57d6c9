"""GB/s of the frequency-split kernels (Blur15, Laplacian3, fused Fusion-D input) -- SURVEY 8(d) depthwise rows."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "fd-gan_amd"))
from fdgan_hip import engine as E
from loss import fusion_input
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (b, s) in ((16, 256), (4, 1024)):
    x = torch.rand(b, 3, s, s, device="cuda")
    byts = 2 * x.numel() * 4                      # fp32 NCHW planes: read once + write once
    tb, tl = t(lambda: E.blur15(x, True)), t(lambda: E.laplacian3(x))
    with torch.no_grad():
        tf = t(lambda: fusion_input(x))
    print(f"B={b} @{s}x{s}: blur15 {tb:7.1f} us {byts/tb/1e3:7.1f} GB/s | laplacian3 {tl:7.1f} us {byts/tl/1e3:7.1f} GB/s | fusion_input (img+LF+HF, 4 planes moved) {tf:7.1f} us {2*byts/tf/1e3:7.1f} GB/s")
