"""Robustness run of the training step (experiment aid): several sizes, many steps, memory must stay flat."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fd-gan_amd")]
import torch
import train as T
dev = torch.device("cuda:0")
for (B, S, n) in ((16, 256, 60), (4, 512, 6), (2, 96, 6), (1, 1024, 3), (16, 256, 3)):
    ts = T.TrainStep(dev)      # (the image pool of a step object keeps images of ONE size, as the reference's does)
    gt = torch.rand(B, 3, S, S, device=dev); haze = (gt * 0.6 + 0.3).clamp(0, 1)
    ts.step(haze, gt); torch.cuda.synchronize()
    m0 = torch.cuda.memory_allocated(); t0 = time.time()
    for i in range(n):
        r = ts.step(haze, gt)
    torch.cuda.synchronize(); dt = (time.time() - t0) / n
    m1 = torch.cuda.memory_allocated()
    ok = all(v == v and abs(v) < 1e6 for v in r.values())
    print(f"B {B} S {S}: {dt*1e3:7.1f} ms/step {B/dt:7.1f} img/s  mem {m0/2**30:.2f} -> {m1/2**30:.2f} GiB (peak {torch.cuda.max_memory_allocated()/2**30:.2f})  finite={ok} lossG={r['lossG']:.4f}")
