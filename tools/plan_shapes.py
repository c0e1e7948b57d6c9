"""Per-conv time of the forward plans of D and VGG16 at the training batch (experiment aid)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fd-gan_amd")]
import torch
import models.dehaze1113 as net
from myutils.vgg16 import Vgg16
dev = torch.device("cuda:0")
for name, m, x in (("D", net.D(9, 36).to(dev), torch.rand(16, 9, 256, 256, device=dev)), ("VGG16", Vgg16().to(dev), torch.rand(16, 3, 256, 256, device=dev))):
    with torch.no_grad():
        for _ in range(3): m(x)
        plan = m.hip_plan(x) if hasattr(m, 'hip_plan') else m._plan_for(x)
        ms = plan.main.profile()
        names = plan.main.kernel_names()
    print(name, "total %.3f ms" % sum(ms))
    for meta in plan.meta:
        if not meta["launches"] or "cin" not in meta: continue
        k0 = meta["launches"][0]
        t = sum(ms[i] for i in meta["launches"])
        print("  %-22s %4d->%4d @%3dx%3d  %8.1f us  %7.1f TFLOP/s  %6.2f TB/s" % (names[k0], meta["cin"], meta["cout"], meta["h_out"], meta["w_out"],
              t * 1e3, meta["flops"] / t / 1e9, meta["bytes"] / t / 1e9))
