"""Times the generator's forward + backward (the part of the training step that exists) at the benchmark size."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fd-gan_amd")]
import torch
import models.dehaze1113 as net
B, S = int(sys.argv[1]) if len(sys.argv) > 1 else 16, int(sys.argv[2]) if len(sys.argv) > 2 else 256
torch.manual_seed(0)
g = net.FDGAN().to("cuda:0")
x = torch.rand(B, 3, S, S, device="cuda:0"); tgt = torch.rand(B, 3, S, S, device="cuda:0") * 2 - 1
def step():
    g.zero_grad(set_to_none=True)
    y = g(x)
    ((y - tgt) ** 2).mean().backward()
for _ in range(2): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 3
for _ in range(n): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
with torch.no_grad():
    for _ in range(2): g(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): g(x)
    torch.cuda.synchronize(); df = (time.perf_counter() - t0) / 5
print({"batch": B, "size": S, "fwd_ms": round(df * 1e3, 2), "fwd_bwd_ms": round(dt * 1e3, 2), "images_per_s_fwd_bwd": round(B / dt, 1),
       "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2**30, 2)})
