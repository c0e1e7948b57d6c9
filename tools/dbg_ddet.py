"""Debug aid: run-to-run determinism of the discriminator's parameter gradients at B=16 @ 256^2 (three fresh TrainStep
objects from the same seed).  Modes: step | real | real_side | both_main | twice_main"""
import sys, os
sys.path[:0] = [os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."), os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fd-gan_amd")]
import torch, numpy as np
import train
from loss import fusion_input
from fdgan_hip.losses import bce_loss
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(21)
gt = torch.rand(16, 3, 256, 256, generator=g).to(dev)
haze = (gt * 0.6 + 0.3).clamp(0, 1)
mode = sys.argv[1] if len(sys.argv) > 1 else "step"
res = []
for k in range(3):
    torch.manual_seed(1234); np.random.seed(99)
    ts = train.TrainStep(dev, synthetic=True)
    def real():
        with torch.no_grad():
            real_in = fusion_input(gt)
        bce_loss(ts.netD(real_in), 1.0).backward()
    if mode == "step":
        ts.step(haze, gt)
    else:
        ts._set_d_grad(True); ts.optD.zero_grad()
        main = torch.cuda.current_stream(dev)
        if mode == "real":
            real()
        elif mode == "real_side":
            ts.side.wait_stream(main)
            with torch.cuda.stream(ts.side):
                real()
            with torch.no_grad():
                fake = ts.netG(haze)
            main.wait_stream(ts.side)
        elif mode in ("both_main", "twice_main"):
            real()
            with torch.no_grad():
                fake = ts.netG(haze) if mode == "both_main" else gt
                fake_in = fusion_input(fake)
            bce_loss(ts.netD(fake_in), 0.0).backward()
    torch.cuda.synchronize()
    res.append({n: p.grad.detach().clone() for n, p in ts.netD.named_parameters()})
    del ts
print("mode", mode)
for n in res[0]:
    d1 = float((res[0][n] - res[1][n]).abs().max()); d2 = float((res[1][n] - res[2][n]).abs().max())
    if d1 or d2:
        print("%-40s max|g| %.3e  diff01 %.3e diff12 %.3e" % (n, float(res[0][n].abs().max()), d1, d2))
