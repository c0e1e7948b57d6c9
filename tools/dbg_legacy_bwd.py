"""Debug aid: per-parameter gradient error of a legacy network's HIP backward vs the functional oracle, in reverse network order.
    python tools/dbg_legacy_bwd.py dense1113|dense2_1113|dense22|G|G2|dehaze"""
import importlib, os, sys, torch
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [R, os.path.join(R, "fd-gan_amd"), os.path.join(R, "tests")]
from oracle import legacy_ref
from oracle.detweights import det_input, fill_state_dict
which = sys.argv[1]
DEV = "cuda:0"
torch.manual_seed(0)
if which.startswith("dense"):
    mod, cls, tail = {"dense1113": ("dehaze1113", "Dense", "bn"), "dense2_1113": ("dehaze1113", "Dense2", "pyramid"), "dense22": ("dehaze22", "Dense", "pyramid")}[which]
    net = getattr(importlib.import_module("models." + mod), cls)()
    fill_state_dict(net, seed=6)
    with torch.no_grad():
        net.refine3.weight.mul_(0.1), net.refine3.bias.mul_(0.1)
    x = det_input((2, 3, int(os.environ.get("H", "64")), int(os.environ.get("W", "96"))), seed=33)
    fwd = lambda sdg: legacy_ref.dense_forward(sdg, x.clone(), True, tail)
    masks = None
elif which in ("G", "G2"):
    import models.dehaze22 as net22
    net = getattr(net22, which)(3, 3, 8)
    fill_state_dict(net, seed=5)
    x = det_input((2, 3, 256, 256), seed=21)
    masks = [(torch.rand(2, 64) > 0.5).float() * 2.0 for _ in range(3)]
    fwd = lambda sdg: legacy_ref.unet_forward(sdg, x.clone(), True, which, masks=list(masks))[0]
sd = {k: v.clone() for k, v in net.state_dict().items()}
cot = det_input(tuple(x.shape), seed=7, lo=-1.0, hi=1.0)
sdg = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running_" not in k else v.clone()) for k, v in sd.items()}
(fwd(sdg) * cot).sum().backward()
net = net.to(DEV).train()
if masks is not None:
    net.__dict__["_forced_dropout_masks"] = [m.to(DEV) for m in masks]
y = net(x.to(DEV))
(y * cot.to(DEV)).sum().backward()
torch.cuda.synchronize()
rows = []
for k, p in net.named_parameters():
    g = sdg[k].grad
    if g is None:
        rows.append((k, None, None, None if p.grad is None else float(p.grad.norm())))
        continue
    h = p.grad.cpu() if p.grad is not None else torch.zeros_like(g)
    rows.append((k, float((h - g).norm() / (g.norm() + 1e-30)), float(h.norm() / (g.norm() + 1e-30)), float(g.norm())))
for k, e, ratio, n in reversed(rows):
    print("%-55s err %-10s ratio %-8s |gref| %s" % (k, "-" if e is None else "%.4f" % e, "-" if ratio is None else "%.3f" % ratio, "-" if n is None else "%.3e" % n))
if os.environ.get("CHECKS"):
    from hiputil import op_reference
    from models.dehaze1113 import _plan_backward
    B = _plan_backward(net._plan_for(x.to(DEV)))
    B.checks, B.check_reference = [], op_reference
    net.zero_grad()
    (net(x.to(DEV)) * cot.to(DEV)).sum().backward()
    torch.cuda.synchronize()
    for o in B.checks:
        print("%-34s dw %.4f dx %s" % (o["label"], o["dw"], "%.4f" % o["dx"] if "dx" in o else "-"))
