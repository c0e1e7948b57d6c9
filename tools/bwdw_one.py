"""Time fdgan_conv1x1_bwd_data_weight (the fused bottleneck backward) on the generator's own shapes (experiment aid):
    python tools/bwdw_one.py            every (size, C) of the three dense blocks at B = 16, accumulate + dy_affine as the step runs it
    python tools/bwdw_one.py 256 224    one shape"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fd-gan_amd"))
from fdgan_hip import engine as E, lib as L
N, dev = 16, "cuda"
shapes = [(int(sys.argv[1]), int(sys.argv[2]))] if len(sys.argv) > 2 else \
    [(256, 64 + 32 * k) for k in range(6)] + [(128, 128 + 32 * k) for k in range(12)] + [(64, 256 + 32 * k) for k in range(24)]
ACC = int(os.environ.get("ACC", "1"))
AFF = int(os.environ.get("AFF", "1"))
tot_t = tot_b = 0.0
for hw, c in shapes:
    pitch = (c + 127) // 128 * 128
    x = torch.randn(N, hw, hw, pitch, device=dev).half()
    G = torch.zeros(N, hw, hw, pitch, device=dev).bfloat16()
    dy = (torch.randn(N, hw, hw, 128, device=dev) * 0.1).bfloat16()
    yb = torch.randn(N, hw, hw, 128, device=dev).half()
    w = torch.randn(128, c, 1, 1, device=dev) * 0.05
    pw = E.PackedWeight(w, c, 128, 1, transposed=False, flip=True, stride=1, layout=L.WLAYOUT_CHUNK32)
    pw.pack()
    keep = [torch.zeros(c, device=dev), torch.ones(c, device=dev), torch.ones(c, device=dev), torch.zeros(c, device=dev)]
    pro = E.make_prologue(mean=keep[0], var=keep[1], gamma=keep[2], beta=keep[3], act=1)
    ws_bn, ws = torch.empty(1 << 22, device=dev), torch.empty(1 << 26, device=dev)
    dw = torch.zeros(128, c, device=dev)
    cB, cC = torch.randn(128, device=dev) * 0.01, torch.randn(128, device=dev) * 0.01
    xv, gv, dv, ybv = E.View(x, 0, c), E.View(G, 0, c), E.View(dy), E.View(yb)
    aff = (ybv.fd, cB, cC) if AFF else None
    run = lambda: E.conv1x1_bwd_data_weight(dv.fd, pw, xv.fd, pro, gv.fd, ws_bn, ACC, ws, dw, True, dy_affine=aff)
    for _ in range(2):
        assert run() is not None
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(5):
        run()
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 5 * 1e3 - 7.0          # minus the wgrad_reduce launch that follows each call (~7 us)
    P = N * hw * hw
    byts = P * (128 * (2 if AFF else 1) + (3 if ACC == 1 else 2) * c) * 2
    tot_t += t
    tot_b += byts
    print(f"hw {hw:4d} C {c:4d}: {t:7.1f} us  {byts / t / 1e6:5.2f} TB/s", flush=True)
    del x, G, dy, yb
print(f"sum {tot_t / 1e3:.3f} ms  {tot_b / tot_t / 1e6:.2f} TB/s  acc={ACC} affine={AFF}")
