"""Debug aid: bitwise reproducibility of one weight gradient while ANOTHER stream keeps the GPU busy (timing-dependent races)."""
import sys, os
sys.path[:0] = [os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."), os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fd-gan_amd")]
import torch
from fdgan_hip import engine as E, lib as L
dev = torch.device("cuda:0")
n, cin, cout, h, w, k, pad = 16, 144, 288, 128, 128, 4, 1
if len(sys.argv) > 1:
    n, cin, cout, h, w, k, pad = [int(v) for v in sys.argv[1:8]]
ho, wo = h + 2 * pad - k + 1, w + 2 * pad - k + 1
torch.manual_seed(0)
x = torch.randn(n, h, w, cin, device=dev).to(torch.float16)
dy = (torch.randn(n, ho, wo, cout, device=dev) * 0.1).to(torch.bfloat16)
mean, var = torch.randn(cin, device=dev) * 0.1, torch.rand(cin, device=dev) + 0.5
gamma, beta = torch.rand(cin, device=dev) + 0.5, torch.randn(cin, device=dev) * 0.1
pro = E.make_prologue(act=L.ACT_LEAKY02, mean=mean, var=var, gamma=gamma, beta=beta)
ws = torch.empty(1 << 26, dtype=torch.float32, device=dev)
desc = E.conv_desc(k, 1, pad, cout=cout)
side = torch.cuda.Stream()
a = torch.randn(8192, 8192, device=dev, dtype=torch.float16)
big = torch.randn(1 << 28, device=dev)
# our own data-gradient kernel of the same layer as the neighbour (mode 4): what the training step runs beside the wgrad
wt = torch.randn(cout, cin, k, k, device=dev) * 0.05
pwf = E.PackedWeight(wt, cin, cout, k, transposed=False, flip=True, stride=1, layout=L.WLAYOUT_CHUNK32)
pwf.pack()
G = torch.zeros(n, h, w, cin, device=dev, dtype=torch.bfloat16)
ws_bn = torch.empty(1 << 24, dtype=torch.float32, device=dev)
ddesc = E.conv_desc(k, 1, k - 1 - pad, cout=cin, w_layout=L.WLAYOUT_CHUNK32)
outs = []
for it in range(20):
    mode = it % 5
    with torch.cuda.stream(side):
        if mode == 1:
            for _ in range(3): b = a @ a              # MFMA-heavy neighbour
        elif mode == 2:
            for _ in range(6): big.mul_(1.0001)       # HBM-heavy neighbour
        elif mode == 3:
            for _ in range(200): a[:64].add_(1.0)     # many tiny launches
        elif mode == 4:
            for _ in range(2): E.conv_bwd_data(E.View(dy).fd, pwf, E.View(x).fd, pro, E.View(G).fd, ddesc, ws_bn, accumulate=1)
    dw = torch.zeros(cout, cin, k, k, device=dev)
    E.conv_bwd_weight(E.View(x).fd, pro, E.View(dy).fd, desc, dw, None, ws, False)
    torch.cuda.synchronize()
    outs.append(dw.clone())
ref = outs[0]
print("max diffs vs run 0:", ["%.1e" % float((o - ref).abs().max()) for o in outs])
