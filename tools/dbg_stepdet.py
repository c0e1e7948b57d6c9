"""Debug aid: which parameters differ after ONE full-size training step between fresh TrainStep objects built from the same seed."""
import sys, os
sys.path[:0] = [os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."), os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fd-gan_amd")]
import torch, numpy as np
import train
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(21)
gt = torch.rand(16, 3, 256, 256, generator=g).to(dev)
haze = (gt * 0.6 + 0.3).clamp(0, 1)
res = []
for k in range(3):
    torch.manual_seed(1234); np.random.seed(99)
    ts = train.TrainStep(dev, synthetic=True)
    if k == 0 and os.environ.get('ARM'):
        from fdgan_hip import engine as E
        E.kernel_timer_arm(None, 1, 4096)
    ts.step(haze, gt)
    torch.cuda.synchronize()
    res.append({("G." + n): p.grad.detach().clone() for n, p in ts.netG.named_parameters() if p.grad is not None} |
               {("D." + n): p.grad.detach().clone() for n, p in ts.netD.named_parameters() if p.grad is not None})
    del ts
bad = 0
for n in res[0]:
    d1 = float((res[0][n] - res[1][n]).abs().max()); d2 = float((res[1][n] - res[2][n]).abs().max())
    if d1 or d2:
        bad += 1
        print("%-50s max|g| %.3e  diff01 %.3e diff12 %.3e" % (n, float(res[0][n].abs().max()), d1, d2))
print("parameters with run-to-run differences:", bad, "of", len(res[0]))
