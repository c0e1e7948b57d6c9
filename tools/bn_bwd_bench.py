"""Time bn_act_bwd / bn_bwd_apply at the dense-block shapes of the training step (experiment aid)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "fd-gan_amd"))
from fdgan_hip import engine as E

def run(P_hw, C, Ct, N=16):
    dev = "cuda"
    x = torch.randn(N, P_hw, P_hw, Ct, device=dev).half()
    d = torch.randn(N, P_hw, P_hw, Ct, device=dev).bfloat16()
    g = torch.zeros(N, P_hw, P_hw, Ct, device=dev).bfloat16()
    xv, dv, gv = E.View(x, 0, C), E.View(d, 0, C), E.View(g, 0, C)
    mean = torch.zeros(C, device=dev); var = torch.ones(C, device=dev)
    gamma = torch.ones(C, device=dev); beta = torch.zeros(C, device=dev)
    pro = E.make_prologue(mean=mean, var=var, gamma=gamma, beta=beta, act=1)
    ws = torch.empty(4096 * 1024 * 2, device=dev)
    dg = torch.zeros(C, device=dev); db = torch.zeros(C, device=dev)
    def t(fn, n=10):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    Pn = N * P_hw * P_hw
    ta = t(lambda: E.bn_act_bwd(dv.fd, xv.fd, pro, ws))
    tp = t(lambda: E.bn_bwd_apply(dv.fd, xv.fd, pro, dg, db, gv.fd, True))
    print(f"hw {P_hw} C {C}/{Ct}: act_bwd {ta:7.1f} us {6*Pn*C/ta/1e6:6.2f} TB/s | apply {tp:7.1f} us {8*Pn*C/tp/1e6:6.2f} TB/s")

for hw, C, Ct in [(127, 288, 288), (128, 144, 144), (128, 72, 72), (256, 224, 256), (256, 128, 128), (128, 480, 512), (64, 992, 1024), (256, 64, 256)]:
    run(hw, C, Ct)
