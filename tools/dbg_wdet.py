"""Debug aid: bitwise run-to-run determinism of fdgan_conv2d_bwd_weight on one shape (default: D's 4x4 144 -> 288 @ 127)."""
import sys, os
sys.path[:0] = [os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."), os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fd-gan_amd")]
import torch
from fdgan_hip import engine as E, lib as L
dev = torch.device("cuda:0")
n, cin, cout, h, w, k, pad = 16, 144, 288, 127, 127, 4, 1
if len(sys.argv) > 1:
    n, cin, cout, h, w, k, pad = [int(v) for v in sys.argv[1:8]]
ho, wo = h + 2 * pad - k + 1, w + 2 * pad - k + 1
torch.manual_seed(0)
x = (torch.randn(n, h, w, cin, device=dev)).to(torch.float16)
dy = (torch.randn(n, ho, wo, cout, device=dev) * 0.1).to(torch.bfloat16)
mean, var = torch.randn(cin, device=dev) * 0.1, torch.rand(cin, device=dev) + 0.5
gamma, beta = torch.rand(cin, device=dev) + 0.5, torch.randn(cin, device=dev) * 0.1
pro = E.make_prologue(act=L.ACT_LEAKY02, mean=mean, var=var, gamma=gamma, beta=beta)
ws = torch.empty(1 << 26, dtype=torch.float32, device=dev)
desc = E.conv_desc(k, 1, pad, cout=cout)
outs = []
for it in range(12):
    if it % 3 == 1: ws.fill_(float("nan") if it > 6 else 123.0)     # poison: a partial that is read but never written shows up
    dw = torch.zeros(cout, cin, k, k, device=dev)
    E.conv_bwd_weight(E.View(x).fd, pro, E.View(dy).fd, desc, dw, None, ws, False)
    torch.cuda.synchronize()
    outs.append(dw.clone())
ref = outs[0]
print("finite", bool(torch.isfinite(torch.stack(outs)).all()))
print("max diffs vs run 0:", ["%.2e" % float((o - ref).abs().max()) for o in outs])
