"""Time fdgan_conv2d_bwd_data on one dense-layer bottleneck shape (experiment aid): bwd_data_one.py HW C [CT] [Cy]"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "fd-gan_amd"))
from fdgan_hip import engine as E, lib as L
hw, c = int(sys.argv[1]), int(sys.argv[2])
ct = int(sys.argv[3]) if len(sys.argv) > 3 else c
cy = int(sys.argv[4]) if len(sys.argv) > 4 else 128
ks = int(sys.argv[5]) if len(sys.argv) > 5 else 1
N, dev = 16, "cuda"
x = torch.randn(N, hw, hw, ct, device=dev).half()
G = torch.zeros(N, hw, hw, ct, device=dev).bfloat16()
dy = torch.randn(N, hw, hw, cy, device=dev).bfloat16()
w = torch.randn(cy, c, ks, ks, device=dev) * 0.05
pw = E.PackedWeight(w, c, cy, ks, transposed=False, flip=True, stride=1, layout=L.WLAYOUT_CHUNK32)
pw.pack()
mean = torch.zeros(c, device=dev); var = torch.ones(c, device=dev); gamma = torch.ones(c, device=dev); beta = torch.zeros(c, device=dev)
pro = E.make_prologue(mean=mean, var=var, gamma=gamma, beta=beta, act=1)
ws = torch.empty(1 << 24, device=dev)
desc = E.conv_desc(ks, 1, ks // 2, cout=c, w_layout=L.WLAYOUT_CHUNK32)
xv, gv, dv = E.View(x, 0, c), E.View(G, 0, c), E.View(dy, 0, cy)
ACC = int(os.environ.get('ACC', '1'))
def run(): E.conv_bwd_data(dv.fd, pw, xv.fd, pro, gv.fd, desc, ws, accumulate=ACC)
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 10 * 1e3
P = N * hw * hw
byts = P * (cy + (3 if ACC == 1 else 2) * c) * 2
print(f"hw {hw} C {c}/{ct} Cy {cy} k{ks}: {t:8.1f} us  {byts/t/1e6:6.2f} TB/s  {2.0*P*c*cy*ks*ks/t/1e6:7.1f} TFLOP/s  acc={ACC} stream={'off' if (os.environ.get('FDGAN_DEBUG_NO_BWD1X1S') or os.environ.get('FDGAN_DEBUG_NO_BWD3X3S')) else 'on'}")
