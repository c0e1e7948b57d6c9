import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fd-gan_amd"), os.path.join(ROOT, "tests")]
import torch
import models.dehaze1113 as net
from oracle import dehaze1113_ref as ref
from oracle.detweights import det_input, fill_state_dict
from hiputil import rel_rms
od = ref.D(9, 36); fill_state_dict(od, seed=1)
d = net.D(9, 36); d.load_state_dict(od.state_dict()); d = d.to("cuda:0")
x = det_input((2, 9, 64, 64), seed=77, lo=-1.0, hi=1.0)
cot = det_input((2, 1, 30, 30), seed=5, lo=-1.0, hi=1.0)
outs = {}
for mod in od.modules():
    if isinstance(mod, torch.nn.LeakyReLU): mod.inplace = False
m = od.main
convs = [m.layer1.conv, m.layer2.layer2.conv, m.layer3.layer3.conv, m.layer4.conv]
convs.append(m.layer5.conv)
st = lambda t: t + (t.to(torch.bfloat16).float() - t).detach()      # straight-through bf16 rounding
with torch.no_grad():
    for c in convs:
        c.weight.copy_(c.weight.to(torch.bfloat16).float())         # the filter as the kernels see it
d.load_state_dict(od.state_dict())
for c in convs:
    c.register_forward_pre_hook(lambda mod, inp: (st(inp[0]),))     # activated operand rounded as the staging does
for i, c in enumerate(convs[:4]):
    def hook(mod, inp, out, i=i):
        out = out + (out.to(torch.bfloat16).float() - out).detach()     # store-rounding emulation (straight-through)
        out.retain_grad(); outs[i] = out
        return out
    c.register_forward_hook(hook)
xo = x.clone().requires_grad_(True)
(od(xo) * cot).sum().backward()
xg = x.to("cuda:0").requires_grad_(True)
y = d(xg); (y * cot.to("cuda:0")).sum().backward(); torch.cuda.synchronize()
P = d._plan_for(xg)
for i, (dg, a) in enumerate(zip(d._last_act_grads, P.acts)):
    c = outs[i].shape[1]
    g_hip = dg[..., :c].permute(0, 3, 1, 2).float().cpu()
    a_hip = a[..., :c].permute(0, 3, 1, 2).float().cpu()
    print("layer%d: act rel_rms %.4f  grad rel_rms %.4f  |grad| %.3e" % (i + 1, rel_rms(a_hip, outs[i].detach()), rel_rms(g_hip, outs[i].grad), float(outs[i].grad.abs().mean())))

for (k, p), (_, q) in zip(d.named_parameters(), od.named_parameters()):
    print(k, "%.4f" % rel_rms(p.grad.cpu(), q.grad))
print("dx %.4f" % rel_rms(xg.grad.cpu(), xo.grad))
# isolate the first data gradient: d4 = lrelu'(a4) * conv_transpose(g5, W5) from the HIP path's own tensors
import torch.nn.functional as F
a4 = P.acts[3][..., :288].permute(0, 3, 1, 2).float().cpu()
s = y.detach().cpu()
g5 = (cot * s * (1 - s)).to(torch.bfloat16).float()
W5 = od.main.layer5.conv.weight.detach()
da4 = F.conv_transpose2d(g5, W5, None, 1, 1)
ref_d4 = da4 * torch.where(a4 > 0, torch.ones_like(a4), torch.full_like(a4, 0.2))
hip_d4 = d._last_act_grads[3][..., :288].permute(0, 3, 1, 2).float().cpu()
print("d4 vs own-tensor reference: %.4f   (oracle d4 vs that reference: %.4f)" % (rel_rms(hip_d4, ref_d4), rel_rms(outs[3].grad, ref_d4)))
lr = torch.where(outs[3].detach() > 0, 1.0, 0.2); lh = torch.where(a4 > 0, 1.0, 0.2)
print("mask mismatches: %d of %d" % (int((lr != lh).sum()), lr.numel()))
