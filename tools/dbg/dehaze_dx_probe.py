"""dehaze22.dehaze's input-image gradient, output by output, against the fp32 oracle (debug aid)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fd-gan_amd"), os.path.join(ROOT, "tests")]
import torch
import models.dehaze22 as net22
from oracle import legacy_ref
from oracle.detweights import det_input, fill_state_dict
from hiputil import rel_rms, emulated_functional_convs
DEV = torch.device("cuda:0")
net = net22.dehaze(3, 3, 64)
fill_state_dict(net, seed=8)
with torch.no_grad():
    net.tran_dense.refine3.weight.mul_(0.05), net.tran_dense.refine3.bias.fill_(1.0), net.refine3.weight.mul_(0.02)
sd = {k: v.clone() for k, v in net.state_dict().items()}
net = net.to(DEV).train()
x = det_input((4, 3, 256, 256), seed=41)
torch.manual_seed(4)
masks = [(torch.rand(4, 64) > 0.5).float() * 2.0 for _ in range(3)]
net.atp_est.__dict__["_forced_dropout_masks"] = [m.to(DEV) for m in masks]
torch.set_num_threads(16)
for i in range(4):
    cot = det_input((4, 3, 256, 256), seed=50 + i, lo=-1.0, hi=1.0)
    xr = x.clone().requires_grad_(True)
    sdg = {k: v.clone() for k, v in sd.items()}
    outs = legacy_ref.dehaze_forward(sdg, xr.clone(), True, list(masks))[:4]
    (outs[i] * cot).sum().backward()
    xe = x.clone().requires_grad_(True)
    with emulated_functional_convs(legacy_ref):
        outs_e = legacy_ref.dehaze_forward({k: v.clone() for k, v in sd.items()}, xe.clone(), True, list(masks))[:4]
        (outs_e[i] * cot).sum().backward()
    ge = xe.grad.double()
    print("   emulated rounding vs oracle: rel_rms %.4f cosine %.5f" % (rel_rms(ge, xr.grad.double()), float((ge * xr.grad.double()).sum() / (ge.norm() * xr.grad.double().norm()))), flush=True)
    xg = x.to(DEV).requires_grad_(True)
    ys = net(xg)
    (ys[i] * cot.to(DEV)).sum().backward()
    torch.cuda.synchronize()
    gx, gr = xg.grad.double().cpu(), xr.grad.double()
    print("   hip vs emulated: rel_rms %.4f" % rel_rms(xg.grad.double().cpu(), ge))
    print("output %d: rel_rms %.4f cosine %.5f  |oracle| %.4e |hip| %.4e  forward rel_rms %.2e" % (
        i, rel_rms(gx, gr), float((gx * gr).sum() / (gx.norm() * gr.norm())), float(gr.norm()), float(gx.norm()), rel_rms(ys[i].detach().cpu().double(), outs[i].detach().double())), flush=True)
