"""Does the gradient a BatchNorm backward leaves in G sum to zero per channel, as it does analytically?  (debug aid, round 4:
the legacy networks' well-conditioned gradient check found rank-1 errors in the weight gradients of the convs that READ such a
gradient -- the signature of a per-channel DC offset.)  One fused data-gradient launch + finalize_coef + flush on random data."""
import os, sys, torch
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [R, os.path.join(R, "fd-gan_amd"), os.path.join(R, "tests")]
from fdgan_hip import engine as E, lib as L
from hiputil import seeded, f16_round, bf16_round
DEV = "cuda:0"


def nhwc(t, dtype, pitch=None):
    n, c, h, w = t.shape
    pitch = pitch or (c + 7) // 8 * 8
    buf = torch.zeros((n, h, w, pitch), dtype=dtype, device=DEV)
    buf[..., :c] = t.permute(0, 2, 3, 1).to(DEV).to(dtype)
    return buf


def run(k, pad, cin, cout, n, h, w, xmean=0.0):
    x = f16_round(seeded((n, cin, h, w), 71, -1.5, 1.5) + xmean)
    dy = bf16_round(seeded((n, cout, h, w), 72, -1.0, 1.0) * 0.01)
    wt = bf16_round(seeded((cout, cin, k, k), 73, -1.0, 1.0) * (2.0 / (cin * k * k)) ** 0.5)
    mean, var = x.double().mean((0, 2, 3)), x.double().var((0, 2, 3), unbiased=False)
    gamma, beta = seeded((cin,), 76, 0.5, 1.5), seeded((cin,), 77, -0.3, 0.3) + 3.0
    rstd = 1.0 / torch.sqrt(var + 1e-5)
    xhat = (x.double() - mean.view(1, -1, 1, 1)) * rstd.view(1, -1, 1, 1)
    pre = xhat * gamma.double().view(1, -1, 1, 1) + beta.double().view(1, -1, 1, 1)
    da = torch.nn.grad.conv2d_input((n, cin, h, w), wt.double(), dy.double(), stride=1, padding=pad)
    v = da * torch.where(pre > 0, torch.ones_like(pre), torch.full_like(pre, 0.2))
    M = n * h * w
    A = (gamma.double() * rstd).view(1, -1, 1, 1)
    dx_ref = A * (v - v.mean((0, 2, 3), keepdim=True) - xhat * (v * xhat).mean((0, 2, 3), keepdim=True))
    keep = [t.float().to(DEV) for t in (mean, var, gamma, beta)]
    pro = E.make_prologue(act=L.ACT_LEAKY02, mean=keep[0], var=keep[1], gamma=keep[2], beta=keep[3], eps=1e-5)
    pw = E.PackedWeight(wt.to(DEV).contiguous(), cin, cout, k, transposed=False, flip=True, stride=1, layout=L.WLAYOUT_CHUNK32)
    pw.pack()
    xb, dyb = nhwc(x, torch.float16), nhwc(dy, torch.bfloat16)
    G = torch.zeros((n, h, w, (cin + 7) // 8 * 8), dtype=torch.bfloat16, device=DEV)
    ws = torch.zeros(1 << 22, dtype=torch.float32, device=DEV)
    xv, gv = E.View(xb, 0, cin), E.View(G, 0, cin)
    rows, cpad = E.conv_bwd_data(E.View(dyb, 0, cout).fd, pw, xv.fd, pro, gv.fd, E.conv_desc(k, 1, k - 1 - pad, cout=cin, w_layout=L.WLAYOUT_CHUNK32), ws, accumulate=1)
    torch.cuda.synchronize()
    G1 = G[..., :cin].permute(0, 3, 1, 2).double().cpu()
    coef = torch.zeros((2, cin), dtype=torch.float32, device=DEV)
    sg, sb = torch.zeros(cin, device=DEV), torch.zeros(cin, device=DEV)
    E.bn_bwd_finalize_coef(ws, rows, cpad, cin, pro, M, coef[0], coef[1], sink_dgamma=sg, sink_dbeta=sb, scratch=torch.zeros(64 * 4096, device=DEV))
    E.affine_accumulate(xv.fd, coef[0], coef[1], gv.fd)
    torch.cuda.synchronize()
    G2 = G[..., :cin].permute(0, 3, 1, 2).double().cpu()
    dbeta_ref, dgamma_ref = v.sum((0, 2, 3)), (v * xhat).sum((0, 2, 3))
    B_ref = (-gamma.double() * rstd * rstd * dgamma_ref / M)
    C_ref = -(gamma.double() * rstd) * dbeta_ref / M - B_ref * mean
    print("%dx%d %d<-%d @%dx%dx%d xmean %.1f: rel-rms dx %.4f | dbeta: kernel vs sum(stored A*v)/A: %.2e | B rel err %.2e  C rel err %.2e" % (
        k, k, cin, cout, n, h, w, xmean, float((G2 - dx_ref).norm() / dx_ref.norm()),
        float(((sb.cpu().double() - (G1 / A).sum((0, 2, 3))).abs() / (G1 / A).abs().sum((0, 2, 3))).max()),
        float(((coef[0].cpu().double() - B_ref).abs() / B_ref.abs()).median()), float(((coef[1].cpu().double() - C_ref).abs() / C_ref.abs()).median())))
    dc = G2.sum((0, 2, 3)) / G2.abs().sum((0, 2, 3))
    dcr = dx_ref.sum((0, 2, 3)) / dx_ref.abs().sum((0, 2, 3))
    print("   sum(dx) / sum|dx| per channel: hip max %.2e median %.2e   (fp64 reference max %.1e; rounding noise would be ~%.1e)" % (
        float(dc.abs().max()), float(dc.abs().median()), float(dcr.abs().max()), 1.1e-3 / M ** 0.5))
    # what a consumer's weight gradient sees: sum_p dx[p] * a[p] with a = something with a large mean
    a = 3.0 + torch.randn(n, 1, h, w, dtype=torch.float64)
    print("   sum(dx * a) rel err with mean(a) = 3: %.3f" % float(((G2 * a).sum((0, 2, 3)) - (dx_ref * a).sum((0, 2, 3))).norm() / (dx_ref * a).sum((0, 2, 3)).norm()))


for xm in (0.0, 2.0):
    run(3, 1, 20, 3, 4, 128, 128, xm)
    run(1, 0, 128, 128, 4, 64, 64, xm)
    run(3, 1, 128, 32, 4, 64, 64, xm)
