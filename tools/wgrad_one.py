"""Time one weight-gradient shape and check it against torch (experiment aid): [STRIDE=2] python tools/wgrad_one.py HW CIN COUT KS [CT]"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "fd-gan_amd"))
from fdgan_hip import engine as E, lib as L
hw, cin, cout, ks = [int(v) for v in sys.argv[1:5]]
ct = int(sys.argv[5]) if len(sys.argv) > 5 else cin
N = 16
dev = "cuda"
torch.manual_seed(0)
x = torch.randn(N, hw, hw, ct, device=dev).half()
pad = ks // 2 if ks == 3 else (1 if ks == 4 else 0)
ST = int(os.environ.get('STRIDE', '1'))
ho = (hw + 2 * pad - ks) // ST + 1
dy = torch.randn(N, ho, ho, cout, device=dev).bfloat16()
xv, dv = E.View(x, 0, cin), E.View(dy, 0, cout)
mean = 0.1 * torch.randn(cin, device=dev); var = 0.5 + torch.rand(cin, device=dev); gamma = 1 + 0.1 * torch.randn(cin, device=dev); beta = 0.1 * torch.randn(cin, device=dev)
ACT = int(os.environ.get('ACT', '1'))     # 1 ReLU, 2 LeakyReLU(0.2)
pro = E.make_prologue(mean=mean, var=var, gamma=gamma, beta=beta, act=ACT)
desc = L.FdConvDesc(ks, ST, pad, 0, 0, cout, 0)
dw = torch.zeros(cout, cin, ks, ks, device=dev)
ws = torch.empty(64 << 20, device=dev)
def run(): E.conv_bwd_weight(xv.fd, pro, dv.fd, desc, dw, None, ws, False)
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 10 * 1e3
fl = 2.0 * N * ho * ho * cin * cout * ks * ks
# reference: a = relu(bn(x)) rounded to bf16 as the kernel stages it, dW by autograd in fp32 on the GPU
sc = gamma / torch.sqrt(var + 1e-5); sh = beta - mean * sc
pre = x[..., :cin].float() * sc + sh
a = (torch.relu(pre) if ACT == 1 else torch.nn.functional.leaky_relu(pre, 0.2)).bfloat16().float().permute(0, 3, 1, 2)
w = torch.zeros(cout, cin, ks, ks, device=dev, requires_grad=True)
y = torch.nn.functional.conv2d(a, w, stride=ST, padding=pad)
y.backward(dy.float().permute(0, 3, 1, 2))
err = ((dw - w.grad).norm() / w.grad.norm()).item()
print(f"hw {hw} {cin}->{cout} k{ks}: {t:8.1f} us  {fl/t/1e6:7.1f} TFLOP/s  rel err {err:.2e}  R3={os.environ.get('FDGAN_DEBUG_WGRAD_R3','-')}")
