"""Debug / timing aid for csrc/conv_c1.hip: the one-filter 4x4 conv and its data gradient vs torch, and their time at D's size."""
import sys, os
sys.path[:0] = [os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."), os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fd-gan_amd"), os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")]
import torch, torch.nn.functional as F
from fdgan_hip import engine as E, lib as L
dev = torch.device("cuda:0")
torch.manual_seed(0)
def t_us(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (n, h, w, cin) in (((16, 127, 127, 288),) if os.environ.get("BIG") else ((1, 8, 8, 32), (2, 31, 31, 288), (1, 40, 70, 64), (16, 127, 127, 288))):
    x = torch.randn(n, h, w, cin, device=dev).half()
    wt = torch.randn(1, cin, 4, 4, device=dev) * 0.05
    pw = E.PackedWeight(wt, 1, cin, 4); pw.pack()
    out = torch.zeros(n, 1, h - 1, w - 1, device=dev)
    pro = E.make_prologue(act=L.ACT_LEAKY02)
    desc = E.conv_desc(4, 1, 1, L.ACT_NONE, False, cout=1)
    fwd = lambda: E.conv2d(E.View(x).fd, pw, None, pro, E.nchw_f32_view(out), desc)
    fwd(); torch.cuda.synchronize()
    a = F.leaky_relu(x.float().permute(0, 3, 1, 2), 0.2).half().float()
    ref = F.conv2d(a, wt.half().float(), None, 1, 1)
    # data gradient
    dy = (torch.randn(n, h - 1, w - 1, 8, device=dev) * 0.1).bfloat16()
    G = torch.zeros(n, h, w, cin, device=dev, dtype=torch.bfloat16)
    pwf = E.PackedWeight(wt, cin, 1, 4, transposed=False, flip=True, stride=1, layout=L.WLAYOUT_CHUNK32); pwf.pack()
    ddesc = E.conv_desc(4, 1, 2, cout=cin, w_layout=L.WLAYOUT_CHUNK32)
    bwd = lambda: E.conv_bwd_data(E.View(dy, 0, 1).fd, pwf, E.View(x).fd, pro, E.View(G).fd, ddesc, None, accumulate=2)
    bwd(); torch.cuda.synchronize()
    xr = x.float().permute(0, 3, 1, 2)
    da = torch.nn.grad.conv2d_input((n, cin, h, w), wt.bfloat16().float(), dy[..., :1].float().permute(0, 3, 1, 2), stride=1, padding=1)
    gref = da * torch.where(xr > 0, torch.ones_like(xr), torch.full_like(xr, 0.2))
    gerr = float((G.float().permute(0, 3, 1, 2) - gref).abs().max() / gref.abs().max())
    print((n, h, w, cin), "fwd max err %.3g (ref max %.3g)  %.1f us   dgrad rel err %.3g  %.1f us" %
          (float((out - ref).abs().max()), float(ref.abs().max()), t_us(fwd), gerr, t_us(bwd)))
