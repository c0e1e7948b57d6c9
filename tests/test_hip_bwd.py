"""GPU parity of the backward primitives (through the C ABI) against torch autograd on CPU in fp64, with operands
rounded where the kernels round them: the forward input x is an fp16 tensor, dy / G are bf16, the recomputed
a = act(bn(x)) is rounded to bf16 (it is multiplied with the bf16 dy), the flipped filter image is bf16."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from hiputil import bf16_round, f16_round, rel_rms, seeded

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def E():
    from fdgan_hip import engine, lib
    lib.load()
    return engine


def _nhwc(t, pitch=None, dev=DEV, dtype=torch.bfloat16):
    """NCHW fp32 -> NHWC device buffer: bf16 for a gradient (the default), fp16 (_nhwc_a) for a forward activation."""
    n, c, h, w = t.shape
    pitch = pitch or (c + 7) // 8 * 8
    buf = torch.zeros((n, h, w, pitch), dtype=dtype, device=dev)
    buf[..., :c] = t.permute(0, 2, 3, 1).to(dev).to(dtype)
    return buf


def _nhwc_a(t, pitch=None, dev=DEV):
    return _nhwc(t, pitch, dev, torch.float16)


def _from_nhwc(buf, c):
    return buf[..., :c].permute(0, 3, 1, 2).float().cpu()


def _bn_params(c, seed):
    return dict(mean=seeded((c,), seed, -0.3, 0.3), var=seeded((c,), seed + 1, 0.5, 1.5),
                gamma=seeded((c,), seed + 2, 0.5, 1.5), beta=seeded((c,), seed + 3, -0.3, 0.3))


@pytest.mark.parametrize("cin,cout,k,stride,pad,bn,slope", [
    (36, 72, 3, 1, 1, False, 0.2), (72, 144, 3, 1, 1, True, 0.2), (144, 40, 4, 1, 1, True, 0.2), (9, 36, 4, 2, 1, False, 1.0),
    (128, 32, 3, 1, 1, True, 0.0), (96, 128, 1, 1, 0, True, 0.0)])
def test_weight_gradient(E, cin, cout, k, stride, pad, bn, slope):
    from fdgan_hip import lib as L
    n, h, w = 2, 13, 18
    x = f16_round(seeded((n, cin, h, w), 1, -1.5, 1.5))
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    dy = bf16_round(seeded((n, cout, ho, wo), 2, -1.0, 1.0))
    p = _bn_params(cin, 10) if bn else None
    act = {0.0: L.ACT_RELU, 0.2: L.ACT_LEAKY02, 1.0: L.ACT_NONE}[slope]
    # reference: a = bf16(act(bn(x))), dW = conv weight gradient in fp64
    a = x.double()
    if bn:
        sc = (p["gamma"] / torch.sqrt(p["var"] + 1e-5)).float()
        sh = (p["beta"] - p["mean"] * sc).float()
        a = (x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)).double()
    a = torch.where(a > 0, a, a * slope)
    a = bf16_round(a.float()).double()
    wref = torch.zeros(cout, cin, k, k, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(a, wref, None, stride, pad)
    y.backward(dy.double())
    xb, dyb = _nhwc_a(x), _nhwc(dy)
    pro = None
    if bn or act != L.ACT_NONE:
        kw = dict(act=act)
        if bn:
            d = {kk: v.to(DEV) for kk, v in p.items()}
            kw.update(mean=d["mean"], var=d["var"], gamma=d["gamma"], beta=d["beta"], eps=1e-5)
        pro = E.make_prologue(**kw)
    dw = torch.full((cout, cin, k, k), 7.0, dtype=torch.float32, device=DEV)
    db = torch.full((cout,), 7.0, dtype=torch.float32, device=DEV)
    E.conv_bwd_weight(E.View(xb, 0, cin).fd, pro, E.View(dyb, 0, cout).fd, E.conv_desc(k, stride, pad, cout=cout), dw, db)
    torch.cuda.synchronize()
    err = rel_rms(dw.cpu().double(), wref.grad)
    assert err < 5e-3, err
    assert rel_rms(db.cpu().double(), dy.double().sum(dim=(0, 2, 3))) < 1e-3


@pytest.mark.parametrize("c,slope,bn", [(72, 0.2, True), (36, 0.2, False), (128, 0.0, True)])
def test_prologue_backward(E, c, slope, bn):
    from fdgan_hip import lib as L
    n, h, w = 3, 11, 14
    x = f16_round(seeded((n, c, h, w), 3, -1.5, 1.5))
    da = bf16_round(seeded((n, c, h, w), 4, -1.0, 1.0))
    act = {0.0: L.ACT_RELU, 0.2: L.ACT_LEAKY02}[slope]
    xr = x.double().requires_grad_(True)
    if bn:
        gamma = seeded((c,), 5, 0.5, 1.5).double().requires_grad_(True)
        beta = seeded((c,), 6, -0.3, 0.3).double().requires_grad_(True)
        pre = F.batch_norm(xr, None, None, gamma, beta, True, 0.1, 1e-5)
        mean = x.double().mean(dim=(0, 2, 3)).float()
        var = x.double().var(dim=(0, 2, 3), unbiased=False).float()
    else:
        pre = xr
    a = torch.where(pre > 0, pre, pre * slope)
    a.backward(da.double())
    xb, dab = _nhwc_a(x, pitch=(c + 7) // 8 * 8 + 8), _nhwc(da)
    xv, dav = E.View(xb, 0, c), E.View(dab, 0, c)
    if bn:
        keep = [mean.to(DEV), var.to(DEV), gamma.detach().float().to(DEV), beta.detach().float().to(DEV)]   # the struct holds raw pointers
        pro = E.make_prologue(act=act, mean=keep[0], var=keep[1], gamma=keep[2], beta=keep[3], eps=1e-5)
        ws = torch.zeros(1 << 20, dtype=torch.float32, device=DEV)
        rows, cpad = E.bn_act_bwd(dav.fd, xv.fd, pro, ws)
        dg, db = torch.zeros(c, device=DEV), torch.zeros(c, device=DEV)
        E.bn_bwd_finalize(ws, rows, cpad, c, dg, db)
        dxb = torch.zeros_like(dab)
        E.bn_bwd_apply(dav.fd, xv.fd, pro, dg, db, E.View(dxb, 0, c).fd)
        torch.cuda.synchronize()
        assert rel_rms(dg.cpu().double(), gamma.grad) < 6e-3
        assert rel_rms(db.cpu().double(), beta.grad) < 6e-3
        assert rel_rms(_from_nhwc(dxb, c).double(), xr.grad) < 1e-2
    else:
        pro = E.make_prologue(act=act)
        E.bn_act_bwd(dav.fd, xv.fd, pro)
        torch.cuda.synchronize()
        assert rel_rms(_from_nhwc(dab, c).double(), xr.grad) < 6e-3


def test_data_gradient_as_flipped_forward_conv_and_direct(E):
    from fdgan_hip import lib as L
    n, cin, cout, h, w = 2, 72, 144, 12, 17
    for (k, pad) in ((3, 1), (4, 1)):
        ho, wo = h + 2 * pad - k + 1, w + 2 * pad - k + 1
        dy = bf16_round(seeded((n, cout, ho, wo), 7, -1.0, 1.0))
        wt = seeded((cout, cin, k, k), 8, -1.0, 1.0) * (2.0 / (cin * k * k)) ** 0.5
        xr = torch.zeros(n, cin, h, w, dtype=torch.float64, requires_grad=True)
        F.conv2d(xr, bf16_round(wt).double(), None, 1, pad).backward(dy.double())
        dyb = _nhwc(dy)
        wd = wt.to(DEV).contiguous()
        pw = E.PackedWeight(wd, cin, cout, k, transposed=False, flip=True, stride=1, layout=L.WLAYOUT_CHUNK32)
        pw.pack()
        dxb = torch.zeros((n, h, w, cin), dtype=torch.bfloat16, device=DEV)
        E.conv2d(E.View(dyb, 0, cout).fd, pw, None, None, E.View(dxb, 0, cin).fd,
                 E.conv_desc(k, 1, k - 1 - pad, cout=cin, w_layout=L.WLAYOUT_CHUNK32))
        torch.cuda.synchronize()
        assert rel_rms(_from_nhwc(dxb, cin).double(), xr.grad) < 6e-3, (k, pad)
    # strided 4x4 (D's first conv): direct kernel into NCHW fp32
    cin, cout, k, s, pad, h, w = 9, 36, 4, 2, 1, 20, 26
    ho, wo = (h + 2 * pad - k) // s + 1, (w + 2 * pad - k) // s + 1
    dy = bf16_round(seeded((n, cout, ho, wo), 9, -1.0, 1.0))
    wt = seeded((cout, cin, k, k), 10, -0.3, 0.3)
    xr = torch.zeros(n, cin, h, w, dtype=torch.float64, requires_grad=True)
    F.conv2d(xr, f16_round(wt).double(), None, s, pad).backward(dy.double())      # the forward's fp16 filter
    dx = torch.full((n, cin, h, w), 5.0, dtype=torch.float32, device=DEV)
    dyv, wdev = E.View(_nhwc(dy), 0, cout), wt.to(DEV).contiguous()     # keep the buffers alive across the launch
    E.conv_bwd_data_direct(dyv.fd, wdev, E.conv_desc(k, s, pad, cout=cout), dx)
    torch.cuda.synchronize()
    assert rel_rms(dx.cpu().double(), xr.grad) < 1e-5
    # pixel-per-thread variant (3 / 9 / 16 input channels, dy rows padded to whole 8-channel vectors holding junk)
    for (cin, cout, k, s, pad) in ((9, 36, 4, 2, 1), (3, 64, 3, 1, 1), (16, 3, 3, 1, 1)):
        ho, wo = (h + 2 * pad - k) // s + 1, (w + 2 * pad - k) // s + 1
        dy = bf16_round(seeded((n, cout, ho, wo), 11, -1.0, 1.0))
        wt = seeded((cout, cin, k, k), 12, -0.3, 0.3)
        xr = torch.zeros(n, cin, h, w, dtype=torch.float64, requires_grad=True)
        F.conv2d(xr, f16_round(wt).double(), None, s, pad).backward(dy.double())
        dx = torch.full((n, cin, h, w), 5.0, dtype=torch.float32, device=DEV)
        buf = _nhwc(dy, pitch=(cout + 7) // 8 * 8)
        buf[..., cout:] = float("nan")
        dyv, wdev = E.View(buf, 0, cout), wt.to(DEV).contiguous()
        E.conv_bwd_data_direct(dyv.fd, wdev, E.conv_desc(k, s, pad, cout=cout), dx)
        torch.cuda.synchronize()
        assert rel_rms(dx.cpu().double(), xr.grad) < 1e-5, (cin, cout, k, s)


@pytest.mark.parametrize("k,pad,cin,cout,act,dims", [
    (1, 0, 96, 128, "relu", (2, 13, 19)), (3, 1, 128, 32, "relu", (2, 13, 19)), (4, 1, 72, 144, "leaky", (2, 13, 19)),
    (3, 1, 16, 40, "relu", (2, 13, 19)),
    # N*H*W a multiple of 64 and at most 128 forward filters: the streaming 1x1 kernel (conv1x1_bwd.hip)
    (1, 0, 96, 128, "relu", (2, 16, 24)), (1, 0, 224, 128, "relu", (1, 8, 64)), (1, 0, 40, 64, "leaky", (2, 8, 8)),
    (1, 0, 992, 128, "relu", (1, 8, 8)), (3, 1, 128, 32, "relu", (3, 80, 96)),
    # the growth conv's data gradient (32 -> 128): row-streaming kernel (conv3x3_bwd.hip); ragged width, short images
    (3, 1, 128, 32, "relu", (2, 5, 70)), (3, 1, 128, 32, "leaky", (1, 33, 64)),
    # second-generation kernel (widths that are multiples of 64): several column blocks and row segments, rows % 4 in {0, 1, 2}
    (3, 1, 128, 32, "relu", (2, 64, 128)), (3, 1, 128, 32, "relu", (1, 42, 256)), (3, 1, 128, 32, "leaky", (3, 17, 64))])
def test_data_gradient_with_masked_epilogue(E, k, pad, cin, cout, act, dims):
    """fdgan_conv2d_bwd_data: conv^T(dy, W) * act'(bn(x)) stored by the data-gradient kernel itself, with the raw
    moments (sum dpre, sum dpre * x) -> fdgan_bn_bwd_finalize_raw = BatchNorm's (dgamma, dbeta); against torch."""
    from fdgan_hip import lib as L
    n, h, w = dims
    x = f16_round(seeded((n, cin, h, w), 71, -1.5, 1.5))
    ho, wo = h + 2 * pad - k + 1, w + 2 * pad - k + 1
    dy = bf16_round(seeded((n, cout, ho, wo), 72, -1.0, 1.0))
    wt = bf16_round(seeded((cout, cin, k, k), 73, -1.0, 1.0) * (2.0 / (cin * k * k)) ** 0.5)
    p = _bn_params(cin, 74)
    slope = 0.0 if act == "relu" else 0.2
    rstd = 1.0 / torch.sqrt(p["var"].double() + 1e-5)
    xhat = (x.double() - p["mean"].double().view(1, -1, 1, 1)) * rstd.view(1, -1, 1, 1)
    pre = xhat * p["gamma"].double().view(1, -1, 1, 1) + p["beta"].double().view(1, -1, 1, 1)
    da = torch.nn.grad.conv2d_input((n, cin, h, w), wt.double(), dy.double(), stride=1, padding=pad)
    dpre_ref = da * torch.where(pre > 0, torch.ones_like(pre), torch.full_like(pre, slope))
    keep = [v.to(DEV) for v in (p["mean"], p["var"], p["gamma"], p["beta"])]
    pro = E.make_prologue(act=L.ACT_RELU if act == "relu" else L.ACT_LEAKY02, mean=keep[0], var=keep[1], gamma=keep[2], beta=keep[3], eps=1e-5)
    wd = wt.to(DEV).contiguous()
    pw = E.PackedWeight(wd, cin, cout, k, transposed=False, flip=True, stride=1, layout=L.WLAYOUT_CHUNK32)
    pw.pack()
    xb, dyb = _nhwc_a(x, pitch=cin + 8), _nhwc(dy)
    T = torch.full((n, h, w, cin + 8), 7.0, dtype=torch.bfloat16, device=DEV)
    ws = torch.zeros(1 << 20, dtype=torch.float32, device=DEV)
    rows, cpad = E.conv_bwd_data(E.View(dyb, 0, cout).fd, pw, E.View(xb, 0, cin).fd, pro, E.View(T, 0, cin).fd,
                                 E.conv_desc(k, 1, k - 1 - pad, cout=cin, w_layout=L.WLAYOUT_CHUNK32), ws)
    dg, db = torch.empty(cin, device=DEV), torch.empty(cin, device=DEV)
    sink_g, sink_b = torch.full((cin,), 1.0, device=DEV), torch.full((cin,), -1.0, device=DEV)
    E.bn_bwd_finalize_raw(ws, rows, cpad, cin, keep[0], keep[1], 1e-5, dg, db, sink_g, sink_b)
    torch.cuda.synchronize()
    got = _from_nhwc(T, cin).double()
    # a pre-activation within bf16 rounding of zero may fall on the other side of the kink: compare away from it
    safe = pre.abs() > 2e-2
    assert rel_rms(got[safe], dpre_ref[safe]) < 8e-3
    assert float(T[..., cin:].float().min()) == 7.0                       # channels outside the view untouched
    dpre_dev = got                                                        # sums of what the kernel stored (fp32 in-kernel, bf16 stored)
    assert rel_rms(db.cpu().double(), dpre_dev.sum(dim=(0, 2, 3))) < 5e-3
    assert rel_rms(dg.cpu().double(), (dpre_dev * xhat).sum(dim=(0, 2, 3))) < 2e-2
    assert torch.allclose(sink_g.cpu() - 1.0, dg.cpu(), atol=1e-4, rtol=1e-4) and torch.allclose(sink_b.cpu() + 1.0, db.cpu(), atol=1e-4, rtol=1e-4)
    # accumulate mode: the gradient buffer of x receives gamma * rstd * dpre on top of what it holds, dpre is not stored
    g0 = bf16_round(seeded((n, cin, h, w), 75, -0.5, 0.5))
    G = _nhwc(g0, pitch=cin + 8)
    ws.zero_()
    rows2, cpad2 = E.conv_bwd_data(E.View(dyb, 0, cout).fd, pw, E.View(xb, 0, cin).fd, pro, E.View(G, 0, cin).fd,
                                   E.conv_desc(k, 1, k - 1 - pad, cout=cin, w_layout=L.WLAYOUT_CHUNK32), ws, accumulate=True)
    dg2, db2 = torch.empty(cin, device=DEV), torch.empty(cin, device=DEV)
    scratch = torch.zeros(64 * cpad2, dtype=torch.float32, device=DEV)    # two-level reduction when there are > 256 rows
    E.bn_bwd_finalize_raw(ws, rows2, cpad2, cin, keep[0], keep[1], 1e-5, dg2, db2, scratch=scratch)
    torch.cuda.synchronize()
    A = (p["gamma"].double() * rstd).view(1, -1, 1, 1)
    want = g0.double() + A * dpre_ref
    gotG = _from_nhwc(G, cin).double()
    assert rel_rms(gotG[safe], want[safe]) < 8e-3
    assert rel_rms(db2.cpu().double(), db.cpu().double()) < 2e-3 and rel_rms(dg2.cpu().double(), dg.cpu().double()) < 2e-2
    # activation only (no norm): mask by the sign of x, no workspace
    pro1 = E.make_prologue(act=L.ACT_RELU)
    E.conv_bwd_data(E.View(dyb, 0, cout).fd, pw, E.View(xb, 0, cin).fd, pro1, E.View(T, 0, cin).fd,
                    E.conv_desc(k, 1, k - 1 - pad, cout=cin, w_layout=L.WLAYOUT_CHUNK32), None)
    torch.cuda.synchronize()
    assert rel_rms(_from_nhwc(T, cin).double(), da * (x.double() > 0)) < 8e-3


def test_weight_gradient_split_k_pool_and_accumulate(E):
    """Split-K partials + fixed-order reduction, accumulation into an existing gradient, and the pooled
    prologue (transition: BN + ReLU + 2x2 average in front of a 1x1 conv)."""
    from fdgan_hip import lib as L
    n, cin, cout, h, w = 4, 256, 128, 32, 48
    x = f16_round(seeded((n, cin, h, w), 21, -1.5, 1.5))
    p = _bn_params(cin, 30)
    keep = [v.to(DEV) for v in (p["mean"], p["var"], p["gamma"], p["beta"])]
    sc = (p["gamma"] / torch.sqrt(p["var"] + 1e-5)).float()
    sh = (p["beta"] - p["mean"] * sc).float()
    act = torch.relu(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    ws = torch.zeros(1 << 22, dtype=torch.float32, device=DEV)
    for pool in (False, True):
        a = bf16_round(F.avg_pool2d(act, 2) if pool else act).double()
        ho, wo = a.shape[2], a.shape[3]
        dy = bf16_round(seeded((n, cout, ho, wo), 22, -1.0, 1.0))
        wref = torch.zeros(cout, cin, 1, 1, dtype=torch.float64, requires_grad=True)
        F.conv2d(a, wref).backward(dy.double())
        pro = E.make_prologue(act=L.ACT_RELU, pool=pool, mean=keep[0], var=keep[1], gamma=keep[2], beta=keep[3], eps=1e-5)
        xb, dyb = _nhwc_a(x), _nhwc(dy)
        dw = torch.full((cout, cin, 1, 1), 1.0, dtype=torch.float32, device=DEV)
        E.conv_bwd_weight(E.View(xb, 0, cin).fd, pro, E.View(dyb, 0, cout).fd, E.conv_desc(1, 1, 0, cout=cout), dw, None, ws, True)
        torch.cuda.synchronize()
        assert rel_rms(dw.cpu().double() - 1.0, wref.grad) < 5e-3, pool          # accumulated onto the ones
        dw2 = torch.empty_like(dw)
        E.conv_bwd_weight(E.View(xb, 0, cin).fd, pro, E.View(dyb, 0, cout).fd, E.conv_desc(1, 1, 0, cout=cout), dw2, None, ws, False)
        dw3 = torch.empty_like(dw)
        E.conv_bwd_weight(E.View(xb, 0, cin).fd, pro, E.View(dyb, 0, cout).fd, E.conv_desc(1, 1, 0, cout=cout), dw3, None, ws, False)
        torch.cuda.synchronize()
        assert torch.equal(dw2, dw3)                                               # deterministic reduction


@pytest.mark.parametrize("n,cin,h,w,pitch", [(2, 224, 16, 24, 256), (1, 992, 8, 8, 1024), (1, 72, 8, 16, 72), (3, 160, 16, 16, 160)])
def test_weight_gradient_1x1_transpose_read_kernel(E, n, cin, h, w, pitch):
    """conv_wgrad1x1_tr (the dense-layer bottleneck: 1x1, 128 filters, BatchNorm + ReLU prologue, N*H*W a multiple of 64):
    Cin not a multiple of 128, channel slices of a wider buffer, several pixel splits; against torch on identical bf16
    operands."""
    from fdgan_hip import lib as L
    cout = 128
    x = f16_round(seeded((n, cin, h, w), 81, -1.5, 1.5))
    dy = bf16_round(seeded((n, cout, h, w), 82, -1.0, 1.0))
    p = _bn_params(cin, 83)
    keep = [v.to(DEV) for v in (p["mean"], p["var"], p["gamma"], p["beta"])]
    sc = (p["gamma"] / torch.sqrt(p["var"] + 1e-5)).float()
    sh = (p["beta"] - p["mean"] * sc).float()
    a = bf16_round(torch.relu(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))).double()
    wref = torch.zeros(cout, cin, 1, 1, dtype=torch.float64, requires_grad=True)
    F.conv2d(a, wref).backward(dy.double())
    pro = E.make_prologue(act=L.ACT_RELU, mean=keep[0], var=keep[1], gamma=keep[2], beta=keep[3], eps=1e-5)
    xb, dyb = _nhwc_a(x, pitch=pitch), _nhwc(dy)
    if pitch > cin:
        xb[..., cin:] = float("nan")                       # neighbouring channels of the buffer must not leak in
    ws = torch.zeros(1 << 22, dtype=torch.float32, device=DEV)
    dw = torch.full((cout, cin, 1, 1), 0.25, dtype=torch.float32, device=DEV)
    E.conv_bwd_weight(E.View(xb, 0, cin).fd, pro, E.View(dyb, 0, cout).fd, E.conv_desc(1, 1, 0, cout=cout), dw, None, ws, True)
    torch.cuda.synchronize()
    assert rel_rms(dw.cpu().double() - 0.25, wref.grad) < 5e-3
    assert bool(torch.isfinite(dw).all())


def test_gradient_plumbing_kernels(E):
    from fdgan_hip import lib as L
    n, c, h, w = 2, 40, 6, 10
    src = bf16_round(seeded((n, c, h, w), 31, -1, 1))
    dst0 = bf16_round(seeded((n, c, h, w), 32, -1, 1))
    sb, db = _nhwc(src, pitch=48), _nhwc(dst0, pitch=64)
    E.grad_ew(E.GRAD_ADD, E.View(sb, 0, c), E.View(db, 0, c))
    torch.cuda.synchronize()
    assert rel_rms(_from_nhwc(db, c), bf16_round(src + dst0)) < 1e-6
    big = E.new_grad(n, 2 * h, 2 * w, 40, DEV, zero=True)
    E.grad_ew(E.GRAD_UNPOOL, E.View(sb, 0, c), E.View(big, 0, c))
    torch.cuda.synchronize()
    assert rel_rms(_from_nhwc(big, c), bf16_round(F.interpolate(src, scale_factor=2, mode="nearest") * 0.25)) < 1e-6
    small = E.new_grad(n, h // 2, w // 2, 40, DEV, zero=True)
    E.grad_ew(E.GRAD_SUMPOOL, E.View(sb, 0, c), E.View(small, 0, c))
    torch.cuda.synchronize()
    assert rel_rms(_from_nhwc(small, c), bf16_round(F.avg_pool2d(src, 2) * 4)) < 1e-6
    refy = f16_round(torch.relu(seeded((n, c, h, w), 33, -1, 1)))      # a stored forward activation
    rb, ob = _nhwc_a(refy), E.new_grad(n, h, w, 40, DEV, zero=True)
    E.grad_ew(E.GRAD_RELU_MASK, E.View(sb, 0, c), E.View(ob, 0, c), ref=E.View(rb, 0, c))
    torch.cuda.synchronize()
    assert rel_rms(_from_nhwc(ob, c), src * (refy > 0)) < 1e-6
    # output activation backward: tanh image (3 channels) and sigmoid map (1 channel)
    for cc, act, f in ((3, L.ACT_TANH, lambda t: 1 - t * t), (1, L.ACT_SIGMOID, lambda t: t * (1 - t))):
        out = (seeded((n, cc, h, w), 34, -0.9, 0.9) if act == L.ACT_TANH else seeded((n, cc, h, w), 34, 0.05, 0.95))
        dout = seeded((n, cc, h, w), 35, -1, 1)
        g = E.new_grad(n, h, w, 8, DEV)
        od, dd = out.to(DEV).contiguous(), dout.to(DEV).contiguous()
        E.out_act_bwd(dd, od, act, E.View(g))
        torch.cuda.synchronize()
        assert rel_rms(_from_nhwc(g, cc), bf16_round(dout * f(out))) < 1e-6
        assert float(g[..., cc:].float().abs().max()) == 0.0


@pytest.mark.parametrize("n,cin,cout,h,w", [(3, 128, 32, 21, 40), (2, 256, 24, 9, 132), (1, 64, 32, 64, 64), (2, 128, 32, 5, 8),
                                             (2, 256, 32, 9, 132), (1, 128, 32, 70, 64), (1, 128, 32, 3, 100),
                                             (1, 72, 144, 20, 70), (2, 36, 72, 9, 33), (1, 160, 128, 8, 64), (1, 640, 40, 6, 32),
                                             # row-walking kernel (conv_wgrad_r3: W % 64 == 0, Cin % 128 == 0, Cout % 32 == 0):
                                             # several column blocks / row segments / cin slices / cout pairs, odd row counts
                                             (2, 128, 32, 64, 128), (1, 128, 32, 37, 256), (1, 256, 64, 19, 64), (3, 128, 32, 2, 64)])
def test_weight_gradient_3x3_all_taps_kernel(E, n, cin, cout, h, w):
    """conv_wgrad3x3_tr (32 filters, Cin % 128 == 0: transpose-read kernel) and conv_wgrad3x3 (Cout <= 32,
    Cin % 32 == 0), the growth-conv shape 3x3 s1 p1: ragged column blocks, several
    row segments, rows above / below the image, against torch on identical bf16 operands; also equal (up to
    summation order) to the per-tap kernel."""
    import os
    from fdgan_hip import lib as L
    x = f16_round(seeded((n, cin, h, w), 41, -1.5, 1.5))
    dy = bf16_round(seeded((n, cout, h, w), 42, -1.0, 1.0))
    p = _bn_params(cin, 50)
    keep = [v.to(DEV) for v in (p["mean"], p["var"], p["gamma"], p["beta"])]
    sc = (p["gamma"] / torch.sqrt(p["var"] + 1e-5)).float()
    sh = (p["beta"] - p["mean"] * sc).float()
    a = bf16_round(torch.relu(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))).double()
    wref = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(a, wref, None, 1, 1).backward(dy.double())
    pro = E.make_prologue(act=L.ACT_RELU, mean=keep[0], var=keep[1], gamma=keep[2], beta=keep[3], eps=1e-5)
    xb, dyb = _nhwc_a(x), _nhwc(dy, pitch=(cout + 7) // 8 * 8 + 16)
    ws = torch.zeros(1 << 23, dtype=torch.float32, device=DEV)
    dw = torch.empty((cout, cin, 3, 3), dtype=torch.float32, device=DEV)
    E.conv_bwd_weight(E.View(xb, 0, cin).fd, pro, E.View(dyb, 0, cout).fd, E.conv_desc(3, 1, 1, cout=cout), dw, None, ws, False)
    dw_direct = torch.empty_like(dw)
    E.conv_bwd_weight(E.View(xb, 0, cin).fd, pro, E.View(dyb, 0, cout).fd, E.conv_desc(3, 1, 1, cout=cout), dw_direct, None, None, False)
    # with a bias gradient, accumulating onto existing values (conv_refin6 / conv_refine4 of the generator have biases)
    dw_acc, db_acc = torch.full_like(dw, 0.5), torch.full((cout,), -2.0, dtype=torch.float32, device=DEV)
    E.conv_bwd_weight(E.View(xb, 0, cin).fd, pro, E.View(dyb, 0, cout).fd, E.conv_desc(3, 1, 1, cout=cout), dw_acc, db_acc, ws, True)
    torch.cuda.synchronize()
    assert rel_rms(dw.cpu().double(), wref.grad) < 5e-3
    assert rel_rms(dw.cpu(), dw_direct.cpu()) < 1e-4
    assert rel_rms(dw_acc.cpu() - 0.5, dw.cpu()) < 1e-4
    assert rel_rms(db_acc.cpu().double() + 2.0, dy.double().sum(dim=(0, 2, 3))) < 1e-5


@pytest.mark.parametrize("n,cin,cout,h,w", [(2, 144, 64, 13, 70), (1, 144, 288, 6, 131), (1, 288, 32, 9, 20), (2, 288, 1, 12, 67),
                                             (1, 144, 288, 64, 64), (3, 144, 32, 5, 33), (1, 48, 32, 130, 128)])   # + row-walking kernel shapes
def test_weight_gradient_4x4_transpose_read_kernel(E, n, cin, cout, h, w):
    """conv_wgrad4x4_tr (the Fusion-discriminator's 4x4 stride-1 pad-1 conv behind BatchNorm + LeakyReLU(0.2),
    /root/reference/models/dehaze1113.py:200-207): ragged column blocks, rows outside the image, two cin slices."""
    from fdgan_hip import lib as L
    x = f16_round(seeded((n, cin, h, w), 61, -1.5, 1.5))
    ho, wo = h - 1, w - 1
    dy = bf16_round(seeded((n, cout, ho, wo), 62, -1.0, 1.0))
    p = _bn_params(cin, 63)
    keep = [v.to(DEV) for v in (p["mean"], p["var"], p["gamma"], p["beta"])]
    sc = (p["gamma"] / torch.sqrt(p["var"] + 1e-5)).float()
    sh = (p["beta"] - p["mean"] * sc).float()
    a = bf16_round(F.leaky_relu(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1), 0.2)).double()
    wref = torch.zeros(cout, cin, 4, 4, dtype=torch.float64, requires_grad=True)
    F.conv2d(a, wref, None, 1, 1).backward(dy.double())
    pro = E.make_prologue(act=L.ACT_LEAKY02, mean=keep[0], var=keep[1], gamma=keep[2], beta=keep[3], eps=1e-5)
    xb, dyb = _nhwc_a(x), _nhwc(dy, pitch=(cout + 7) // 8 * 8 + 8)
    ws = torch.zeros(1 << 24, dtype=torch.float32, device=DEV)
    dw = torch.empty((cout, cin, 4, 4), dtype=torch.float32, device=DEV)
    E.conv_bwd_weight(E.View(xb, 0, cin).fd, pro, E.View(dyb, 0, cout).fd, E.conv_desc(4, 1, 1, cout=cout), dw, None, ws, False)
    dw_direct = torch.empty_like(dw)
    E.conv_bwd_weight(E.View(xb, 0, cin).fd, pro, E.View(dyb, 0, cout).fd, E.conv_desc(4, 1, 1, cout=cout), dw_direct, None, None, False)
    torch.cuda.synchronize()
    assert rel_rms(dw.cpu().double(), wref.grad) < 5e-3
    # (one filter: the workspace path is the vector-ALU kernel of conv_c1.hip, which multiplies the fp32 activated input -- the
    # matrix-pipe kernels, and `a` above, round it to bf16 first: the two paths differ by that rounding, 2e-3)
    assert rel_rms(dw.cpu(), dw_direct.cpu()) < (1e-4 if cout > 1 else 5e-3)
    if cout == 1:
        a32 = F.leaky_relu(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1), 0.2).double()
        w32 = torch.zeros(cout, cin, 4, 4, dtype=torch.float64, requires_grad=True)
        F.conv2d(a32, w32, None, 1, 1).backward(dy.double())
        assert rel_rms(dw.cpu().double(), w32.grad) < 1e-5       # against the UNROUNDED statement: fp32 accumulation only
        dw_acc = torch.full_like(dw, 3.0)
        E.conv_bwd_weight(E.View(xb, 0, cin).fd, pro, E.View(dyb, 0, cout).fd, E.conv_desc(4, 1, 1, cout=cout), dw_acc, None, ws, True)
        torch.cuda.synchronize()
        assert rel_rms(dw_acc.cpu() - 3.0, dw.cpu()) < 1e-6


@pytest.mark.parametrize("n,cin,cout,h,w,k", [(4, 144, 288, 127, 127, 4), (4, 128, 32, 256, 256, 3), (2, 72, 144, 128, 128, 3),
                                              (4, 224, 128, 128, 128, 1)])
def test_weight_gradient_is_bitwise_reproducible_at_full_size(E, n, cin, cout, h, w, k):
    """Six launches of the same weight gradient at the training step's own image sizes (D's 4x4 144 -> 288 @ 127, the growth
    conv @ 256, D's 72 -> 144 @ 128, a bottleneck @ 128): bitwise equal, with the workspace poisoned in between (a partial
    that is read but never written, or an LDS write that a counted barrier does not cover, shows up here: conv_wgrad_r4's
    step barrier left its row write uncovered in round 2 -- 1 % wrong, differently on every launch, at 127 x 127 only)."""
    from fdgan_hip import lib as L
    pad = {4: 1, 3: 1, 1: 0}[k]
    ho, wo = h + 2 * pad - k + 1, w + 2 * pad - k + 1
    torch.manual_seed(0)
    x = torch.randn(n, h, w, cin, device=DEV).to(torch.float16)
    dy = (torch.randn(n, ho, wo, cout, device=DEV) * 0.1).to(torch.bfloat16)
    keep = [torch.randn(cin, device=DEV) * 0.1, torch.rand(cin, device=DEV) + 0.5, torch.rand(cin, device=DEV) + 0.5,
            torch.randn(cin, device=DEV) * 0.1]
    pro = E.make_prologue(act=L.ACT_LEAKY02 if k == 4 else L.ACT_RELU, mean=keep[0], var=keep[1], gamma=keep[2], beta=keep[3])
    ws = torch.empty(1 << 26, dtype=torch.float32, device=DEV)
    desc = E.conv_desc(k, 1, pad, cout=cout)
    outs = []
    for it in range(6):
        if it % 2 == 1:
            ws.fill_(float("nan"))
        dw = torch.zeros(cout, cin, k, k, device=DEV)
        E.conv_bwd_weight(E.View(x).fd, pro, E.View(dy).fd, desc, dw, None, ws, False)
        torch.cuda.synchronize()
        outs.append(dw)
    assert bool(torch.isfinite(outs[0]).all())
    for o in outs[1:]:
        assert torch.equal(o, outs[0]), float((o - outs[0]).abs().max())
    # and right: against torch on the same operands (fp32 accumulation on the device; a = bf16(act(bn(x))) as the kernel rounds it)
    sc = keep[2] / torch.sqrt(keep[1] + 1e-5)
    a = x.float() * sc + (keep[3] - keep[0] * sc)
    a = torch.where(a > 0, a, a * (0.2 if k == 4 else 0.0)).to(torch.bfloat16).float().permute(0, 3, 1, 2).contiguous()
    wref = torch.zeros(cout, cin, k, k, device=DEV, requires_grad=True)
    F.conv2d(a, wref, None, 1, pad).backward(dy.float().permute(0, 3, 1, 2).contiguous())
    assert rel_rms(outs[0].cpu(), wref.grad.cpu()) < 2e-3


def test_deferred_weight_gradient_reductions_are_bitwise_the_immediate_ones(E):
    """fdgan_conv2d_bwd_weight_job(defer) + ONE fdgan_wgrad_tr_reduce_batch over a table of jobs (what a backward walk does for its
    row-walking weight gradients, include/fdgan_hip.h ABI v9) against the per-conv launch with its own reduction: three shapes in
    one table (the growth conv, D's 4x4 144 -> 288, D's 3x3 72 -> 144), accumulate on and off -- bitwise equal."""
    from fdgan_hip import lib as L
    torch.manual_seed(1)
    cases = [(2, 128, 32, 128, 128, 3, False), (2, 144, 288, 63, 63, 4, True), (2, 72, 144, 64, 64, 3, False)]
    jobs, keep, want, got = [], [], [], []
    for n, cin, cout, h, w, k, acc in cases:
        pad = 1
        ho, wo = h + 2 * pad - k + 1, w + 2 * pad - k + 1
        x = torch.randn(n, h, w, cin, device=DEV).to(torch.float16)
        dy = (torch.randn(n, ho, wo, cout, device=DEV) * 0.1).to(torch.bfloat16)
        pro = E.make_prologue(act=L.ACT_RELU)
        desc = E.conv_desc(k, 1, pad, cout=cout)
        base = torch.randn(cout, cin, k, k, device=DEV) if acc else torch.zeros(cout, cin, k, k, device=DEV)
        ref, out = base.clone(), base.clone()
        ws0 = torch.empty(1 << 25, dtype=torch.float32, device=DEV)
        info = E.conv_bwd_weight_job(E.View(x).fd, pro, E.View(dy).fd, desc, ref, ws0, False, acc)      # immediate: launches its reduction
        assert info is not None, "shape left the row-walking kernels"
        ws1 = torch.full((info.items * info.item_stride,), float("nan"), dtype=torch.float32, device=DEV)   # exactly what the job needs
        job = E.conv_bwd_weight_job(E.View(x).fd, pro, E.View(dy).fd, desc, out, ws1, True, acc)          # deferred: partial sums only
        assert job is not None and job.items == info.items and job.item_stride == info.item_stride
        torch.cuda.synchronize()
        assert torch.equal(out, base)                      # nothing reduced yet
        jobs.append(job)
        keep += [x, dy, ws1, out]
        want.append(ref)
        got.append(out)
    table = E.TrReduceTable(jobs, keep, torch.device(DEV))
    table.launch()
    torch.cuda.synchronize()
    for (n, cin, cout, h, w, k, acc), a, b in zip(cases, want, got):
        assert bool(torch.isfinite(b).all()) and torch.equal(a, b), (cin, cout, k, float((a - b).abs().max()))


def test_streaming_kernels_are_reproducible_beside_a_busy_second_stream(E):
    """The training step runs two HIP streams, so every hand-pipelined kernel shares CUs with another kernel's waves -- among
    them the filter-direct convolutions, which raise their wave priority around the MFMA blocks.  A wave that is held back at
    the wrong moment exposes any hand-counted wait or barrier that does not really cover what it is assumed to cover
    (conv_wgrad_r3 / r4's first fragment reads, round 3).  Each kernel below runs alone, then repeatedly while a second stream
    runs D's 4x4 data gradient (s_setprio, 30 KB of LDS: it co-resides with all of them) and an HBM-bound elementwise kernel:
    results must be bitwise those of the solo run."""
    from fdgan_hip import lib as L
    dev = torch.device(DEV)
    torch.manual_seed(1)
    side = torch.cuda.Stream()
    # the neighbour: data gradient of D's 4x4 144 -> 288 @ 128 (conv4x4_wd144_bwd) + a streaming elementwise op
    nb = dict(x=torch.randn(8, 128, 128, 144, device=dev).half(), dy=(torch.randn(8, 127, 127, 288, device=dev) * 0.1).bfloat16(),
              G=torch.zeros(8, 128, 128, 144, device=dev, dtype=torch.bfloat16), ws=torch.empty(1 << 22, device=dev),
              big=torch.randn(1 << 26, device=dev))
    wt = torch.randn(288, 144, 4, 4, device=dev) * 0.05
    nb["pw"] = E.PackedWeight(wt, 144, 288, 4, transposed=False, flip=True, stride=1, layout=L.WLAYOUT_CHUNK32)
    nb["pw"].pack()
    nb["pro"] = E.make_prologue(act=L.ACT_LEAKY02)
    nb["desc"] = E.conv_desc(4, 1, 2, cout=144, w_layout=L.WLAYOUT_CHUNK32)

    def neighbour():
        with torch.cuda.stream(side):
            for _ in range(2):
                E.conv_bwd_data(E.View(nb["dy"]).fd, nb["pw"], E.View(nb["x"]).fd, nb["pro"], E.View(nb["G"]).fd, nb["desc"], None, accumulate=1)
            nb["big"].mul_(1.0001)

    def bn(c):
        return [torch.randn(c, device=dev) * 0.1, torch.rand(c, device=dev) + 0.5, torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev) * 0.1]

    cases = []
    # weight gradients: D's 4x4 (conv_wgrad_r4), the growth conv (conv_wgrad_r3), 72 -> 144 (first generation)
    for (n, cin, cout, h, w, k, act) in ((8, 144, 288, 128, 128, 4, L.ACT_LEAKY02), (8, 128, 32, 128, 128, 3, L.ACT_RELU), (4, 72, 144, 128, 128, 3, L.ACT_LEAKY02)):
        pad, ho, wo = 1, h + 2 - k + 1, w + 2 - k + 1
        x = torch.randn(n, h, w, cin, device=dev).half()
        dy = (torch.randn(n, ho, wo, cout, device=dev) * 0.1).bfloat16()
        keep = bn(cin)
        pro = E.make_prologue(act=act, mean=keep[0], var=keep[1], gamma=keep[2], beta=keep[3])
        ws = torch.empty(1 << 26, dtype=torch.float32, device=dev)
        desc = E.conv_desc(k, 1, pad, cout=cout)

        def run(x=x, dy=dy, pro=pro, ws=ws, desc=desc, cout=cout, cin=cin, k=k, keep=keep):
            dw = torch.zeros(cout, cin, k, k, device=dev)
            E.conv_bwd_weight(E.View(x).fd, pro, E.View(dy).fd, desc, dw, None, ws, False)
            return dw
        cases.append(("wgrad %dx%d %d->%d" % (k, k, cin, cout), run))
    # forward: growth conv (conv3x3_rs2) and bottleneck (conv1x1_ds) with statistics
    for (cin, cout, k) in ((128, 32, 3), (224, 128, 1)):
        x = torch.randn(8, 128, 128, 256, device=dev).half()
        keep = bn(cin)
        pro = E.make_prologue(act=L.ACT_RELU, mean=keep[0], var=keep[1], gamma=keep[2], beta=keep[3])
        pw = E.PackedWeight(torch.randn(cout, cin, k, k, device=dev) * (2.0 / (cin * k * k)) ** 0.5, cout, cin, k)
        pw.pack()
        desc = E.conv_desc(k, 1, k // 2, cout=cout, w_layout=pw.layout)
        ws = torch.zeros(1 << 20, dtype=torch.float32, device=dev)

        def run(x=x, pro=pro, pw=pw, desc=desc, ws=ws, cin=cin, cout=cout, keep=keep):
            y = torch.empty(8, 128, 128, cout, dtype=torch.float16, device=dev)
            info = E.conv2d(E.View(x, 0, cin).fd, pw, None, pro, E.View(y).fd, desc, ws)
            return torch.cat([y.float().flatten(), ws[:info.stats_rows * info.stats_cpad * 2].clone()])
        cases.append(("fwd %dx%d %d->%d" % (k, k, cin, cout), run))
    # backward data: growth conv (conv3x3_bwd_stream2) and the fused bottleneck backward (conv1x1_bwd_wgrad_stream)
    for kind in ("dgrad3x3", "fused1x1"):
        c = 128 if kind == "dgrad3x3" else 224
        cy = 32 if kind == "dgrad3x3" else 128
        k = 3 if kind == "dgrad3x3" else 1
        x = torch.randn(8, 128, 128, 256, device=dev).half()
        dy = (torch.randn(8, 128, 128, cy, device=dev) * 0.1).bfloat16()
        G0 = (torch.randn(8, 128, 128, 256, device=dev) * 0.1).bfloat16()
        keep = bn(c)
        pro = E.make_prologue(act=L.ACT_RELU, mean=keep[0], var=keep[1], gamma=keep[2], beta=keep[3])
        pwf = E.PackedWeight(torch.randn(cy, c, k, k, device=dev) * 0.05, c, cy, k, transposed=False, flip=True, stride=1, layout=L.WLAYOUT_CHUNK32)
        pwf.pack()
        desc = E.conv_desc(k, 1, k // 2, cout=c, w_layout=L.WLAYOUT_CHUNK32)
        ws_bn, ws = torch.empty(1 << 22, device=dev), torch.empty(1 << 26, device=dev)

        def run(kind=kind, x=x, dy=dy, G0=G0, pro=pro, pwf=pwf, desc=desc, ws_bn=ws_bn, ws=ws, c=c, keep=keep):
            G = G0.clone()
            if kind == "dgrad3x3":
                rows, cpad = E.conv_bwd_data(E.View(dy).fd, pwf, E.View(x, 0, c).fd, pro, E.View(G, 0, c).fd, desc, ws_bn, accumulate=1)
                return torch.cat([G.float().flatten(), ws_bn[:rows * cpad * 2].clone()])
            dw = torch.zeros(128, c, device=dev)
            rows, cpad = E.conv1x1_bwd_data_weight(E.View(dy).fd, pwf, E.View(x, 0, c).fd, pro, E.View(G, 0, c).fd, ws_bn, 1, ws, dw, False)
            return torch.cat([G.float().flatten(), ws_bn[:rows * cpad * 2].clone(), dw.flatten()])
        cases.append((kind, run))
    for name, run in cases:
        solo = run()
        torch.cuda.synchronize()
        assert bool(torch.isfinite(solo).all()), name
        for it in range(6):
            neighbour()
            got = run()
            torch.cuda.synchronize()
            assert torch.equal(got, solo), (name, it, float((got - solo).abs().max()))


def test_flat_adam_matches_torch_adam():
    from fdgan_hip.optim import FlatAdam
    torch.manual_seed(3)
    shapes = [(36, 9, 4, 4), (72,), (5, 3, 3, 3), (1,)]
    ref = [torch.nn.Parameter(torch.randn(s)) for s in shapes]
    mine = [torch.nn.Parameter(p.detach().clone().to(DEV)) for p in ref]
    o_ref = torch.optim.Adam(ref, lr=2e-4, betas=(0.5, 0.999))
    o_hip = FlatAdam(mine, lr=2e-4, betas=(0.5, 0.999))
    for step in range(4):
        o_ref.zero_grad(), o_hip.zero_grad()
        gs = [torch.randn(s) * (0.1 + step) for s in shapes]
        for p, q, g in zip(ref, mine, gs):
            (p * g).sum().backward()
            (q * g.to(DEV)).sum().backward()             # autograd accumulates into the flat gradient view
        o_ref.step(), o_hip.step()
    torch.cuda.synchronize()
    for p, q in zip(ref, mine):
        assert q.data_ptr() >= o_hip.flat.data_ptr() and torch.allclose(q.detach().cpu(), p.detach(), rtol=1e-5, atol=1e-7)
    o_hip.param_groups[0]["lr"] = 0.0
    before = o_hip.flat.clone()
    o_hip.step()
    assert torch.equal(before, o_hip.flat)


@pytest.mark.parametrize("cin,cout,k,stride,pad,bn,slope,n,h,w", [
    (3, 64, 3, 1, 1, False, 1.0, 2, 37, 29),       # FDGAN.conv_refin1 (bias), ragged last pixel tile
    (16, 3, 3, 1, 1, False, 0.0, 3, 24, 40),       # FDGAN.conv_refin3 (bias; ReLU prologue; N = 144 + the bias column)
    (9, 36, 4, 2, 1, False, 1.0, 2, 64, 48),       # D.layer1: 4x4 stride 2
    (12, 20, 3, 1, 1, True, 0.2, 1, 19, 23),       # BatchNorm + LeakyReLU prologue, odd sizes
    (3, 64, 3, 1, 1, False, 1.0, 16, 256, 256)])   # full size: 8192 pixel tiles over 512 persistent workgroups
def test_weight_gradient_few_channel_kernel(E, cin, cout, k, stride, pad, bn, slope, n, h, w):
    """conv_wgrad_small (all taps x all input channels as the N dimension of one GEMM, bias as a column of ones) vs an
    fp64 statement; deterministic; accumulates onto an existing gradient."""
    from fdgan_hip import lib as L
    x = f16_round(seeded((n, cin, h, w), 1, -1.5, 1.5))
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    dy = bf16_round(seeded((n, cout, ho, wo), 2, -1.0, 1.0))
    p = _bn_params(cin, 10) if bn else None
    act = {0.0: L.ACT_RELU, 0.2: L.ACT_LEAKY02, 1.0: L.ACT_NONE}[slope]
    a = x
    if bn:
        sc = (p["gamma"] / torch.sqrt(p["var"] + 1e-5)).float()
        sh = (p["beta"] - p["mean"] * sc).float()
        a = x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
    a = bf16_round(torch.where(a > 0, a, a * slope)).double()
    dev_ref = n * h * w > 200000                      # the fp64 reference of the full-size case runs on the GPU
    wref = torch.zeros(cout, cin, k, k, dtype=torch.float64, requires_grad=True, device=DEV if dev_ref else "cpu")
    F.conv2d(a.to(wref.device), wref, None, stride, pad).backward(dy.double().to(wref.device))
    gref = wref.grad.cpu()
    xb, dyb = _nhwc_a(x), _nhwc(dy)
    pro = None
    if bn or act != L.ACT_NONE:
        kw = dict(act=act)
        if bn:
            d = {kk: v.to(DEV) for kk, v in p.items()}
            kw.update(mean=d["mean"], var=d["var"], gamma=d["gamma"], beta=d["beta"], eps=1e-5)
        pro = E.make_prologue(**kw)
    ws = torch.zeros(1 << 24, dtype=torch.float32, device=DEV)
    desc = E.conv_desc(k, stride, pad, cout=cout)
    dw = torch.full((cout, cin, k, k), 1.0, dtype=torch.float32, device=DEV)
    db = torch.full((cout,), 2.0, dtype=torch.float32, device=DEV)
    E.conv_bwd_weight(E.View(xb, 0, cin).fd, pro, E.View(dyb, 0, cout).fd, desc, dw, db, ws, True)       # accumulate
    dw2, db2 = torch.empty_like(dw), torch.empty_like(db)
    E.conv_bwd_weight(E.View(xb, 0, cin).fd, pro, E.View(dyb, 0, cout).fd, desc, dw2, db2, ws, False)
    dw3 = torch.empty_like(dw)
    E.conv_bwd_weight(E.View(xb, 0, cin).fd, pro, E.View(dyb, 0, cout).fd, desc, dw3, None, ws, False)  # no bias column
    torch.cuda.synchronize()
    assert rel_rms(dw2.cpu().double(), gref) < 2e-3, rel_rms(dw2.cpu().double(), gref)
    assert rel_rms(dw.cpu().double() - 1.0, gref) < 2e-3
    bref = dy.double().sum(dim=(0, 2, 3))
    assert rel_rms(db2.cpu().double(), bref) < 1e-3 and rel_rms(db.cpu().double() - 2.0, bref) < 1e-3
    assert torch.equal(dw2, dw3)
