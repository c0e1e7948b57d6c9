"""Why do some GPU tests return wrong numbers under the guard allocator (tools/dbg/guard_alloc.cpp) although its self-test passes?
One fused 3x3 conv (the shape of tests/test_hip_conv.py::test_conv3x3_persistent_filter_kernel) with checks after every step."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fd-gan_amd"), os.path.join(ROOT, "tests")]
import torch
if os.environ.get("FDGAN_TEST_GUARD_ALLOC"):
    alloc = torch.cuda.memory.CUDAPluggableAllocator(os.environ["FDGAN_TEST_GUARD_ALLOC"], "guard_malloc", "guard_free")
    torch.cuda.memory.change_current_allocator(alloc)
from fdgan_hip import engine as E, lib as L
dev = "cuda"
torch.manual_seed(0)
n, h, w, cin, cout, pitch_in, pitch_out, c0_out = [int(v) for v in sys.argv[1:9]] if len(sys.argv) > 8 else (3, 40, 24, 128, 32, 128, 96, 64)
x = torch.randn(n, cin, h, w)
wt = torch.randn(cout, cin, 3, 3) * 0.05
xbuf = torch.full((n, h, w, pitch_in), 7.0, dtype=torch.float16, device=dev)
xbuf[..., :cin] = x.permute(0, 2, 3, 1).to(dev).to(torch.float16)
torch.cuda.synchronize()
print("xbuf ok:", bool(torch.isfinite(xbuf.float()).all()), "ptr %#x" % xbuf.data_ptr(), "bytes", xbuf.numel() * 2)
ybuf = torch.full((n, h, w, pitch_out), 9.0, dtype=torch.float16, device=dev)
torch.cuda.synchronize()
print("ybuf after full: all 9:", bool((ybuf == 9.0).all()), "ptr %#x" % ybuf.data_ptr(), "bytes", ybuf.numel() * 2)
wparam = wt.to(dev).contiguous()
pw = E.PackedWeight(wparam, cout, cin, 3)
pw.pack()
torch.cuda.synchronize()
print("ybuf after pack: all 9:", bool((ybuf == 9.0).all()), " packed image ptr %#x bytes %d" % (pw.buf.data_ptr(), pw.buf.numel()))
mean, var = torch.randn(cin, device=dev) * 0.1, torch.rand(cin, device=dev) + 0.5
g, bt = torch.rand(cin, device=dev) + 0.5, torch.randn(cin, device=dev) * 0.1
pro = E.make_prologue(act=L.ACT_RELU, mean=mean, var=var, gamma=g, beta=bt)
ws = torch.zeros(1 << 22, dtype=torch.float32, device=dev)
desc = E.conv_desc(3, 1, 1, cout=cout, w_layout=pw.layout)
xv, yv = E.View(xbuf, 0, cin), E.View(ybuf, c0_out, cout)
torch.cuda.synchronize()
print("ybuf before conv: all 9:", bool((ybuf == 9.0).all()))
info = E.conv2d(xv.fd, pw, None, pro, yv.fd, desc, ws)
torch.cuda.synchronize()
print("on the device, after conv: poison channels all 9:", bool((ybuf[..., :c0_out] == 9.0).all()), " kernel:", info and (info.grid_x, info.grid_y, info.lds_bytes))
yf = ybuf.float()
torch.cuda.synchronize()
print("on the device, the fp32 copy: poison channels all 9:", bool((yf[..., :c0_out] == 9.0).all()), "ptr %#x bytes %d (%% 4096 = %d)" % (yf.data_ptr(), yf.numel() * 4, (yf.numel() * 4) % 4096))
yb = yf.cpu()
poison = yb[..., :c0_out]
bad = (poison != 9.0)
print("after conv: poison channels intact:", not bool(bad.any()), " bad elements:", int(bad.sum()))
if bad.any():
    idx = bad.nonzero()
    print("   first bad index", idx[0].tolist(), "last", idx[-1].tolist(), "values", poison[bad][:8].tolist())
    rows = bad.any(-1).any(-1)      # [n, h]
    print("   bad rows per image:", [rows[i].nonzero().flatten().tolist()[:12] for i in range(n)])
out = yb[..., c0_out:c0_out + cout]
print("conv output finite:", bool(torch.isfinite(out).all()), "abs mean %.4f" % float(out.abs().mean()), "zeros:", int((out == 0).sum()), "of", out.numel())
