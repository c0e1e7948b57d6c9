#!/usr/bin/env python
"""CU-partitioned co-scheduling, measured (experiment aid).          python tools/cu_mask_sweep.py [--quick]

The step's HBM-bound chain (dense-layer kernels: MFMA busy 0.12-0.21) and its MFMA-bound family (Fusion-D, VGG16, wide convs:
HBM < 25 %) use complementary resources, but every kernel asks for the whole chip, so two streams time-slice.  This tool measures
  1. where the workgroups of a stream created with hipExtStreamCreateWithCUMask really run (census: XCC / SE / CU of every workgroup),
  2. achieved GB/s (HBM-bound kernels, plus a plain streaming copy as the reference curve) and PFLOP/s (MFMA-bound kernels) vs the
     number of CUs a kernel is given -- persistent kernels get their grid from fdgan_set_cu_budget(n),
  3. an HBM-bound and an MFMA-bound kernel side by side on complementary masks vs the same two on unmasked streams (today) vs one
     after the other.
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fd-gan_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch  # noqa: E402
from fdgan_hip import engine as E  # noqa: E402
from fdgan_hip import lib as L  # noqa: E402

DEV = torch.device("cuda:0")
QUICK = "--quick" in sys.argv


def build_helper():
    so = "/tmp/libcumask.so"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so,
                           os.path.join(ROOT, "tools", "ubench", "cumask.hip")])
    h = C.CDLL(so)
    h.cumask_stream_create.restype = C.c_void_p
    h.cumask_stream_create.argtypes = [C.c_int, C.c_void_p]
    h.cumask_census.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    h.cumask_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_int]
    return h


H = build_helper()
NWORDS = 8          # 256 CUs


def mask_bits(bits):
    m = (C.c_uint32 * NWORDS)()
    for b in bits:
        m[b >> 5] |= 1 << (b & 31)
    return m


def bits_for(n, style, lo=0):
    """n CUs starting at logical slot lo.  'linear': mask bits lo .. lo + n - 1.  'perword': (n / 8) bits of every 32-bit word."""
    if style == "linear":
        return list(range(lo, lo + n))
    per, plo = n // 8, lo // 8
    return [w * 32 + plo + k for w in range(8) for k in range(per)]


_streams = {}


def masked_stream(bits):
    key = tuple(bits)
    if key not in _streams:
        s = H.cumask_stream_create(NWORDS, mask_bits(bits))
        assert s, "hipExtStreamCreateWithCUMask failed"
        _streams[key] = torch.cuda.ExternalStream(s, device=DEV)
    return _streams[key]


def census(stream, nwg=2048, spin=200):
    out = torch.zeros(nwg, 2, dtype=torch.int32, device=DEV)
    torch.cuda.synchronize()
    rc = H.cumask_census(C.c_void_p(stream.cuda_stream), C.c_void_p(out.data_ptr()), nwg, spin)
    assert rc == 0, rc
    torch.cuda.synchronize()
    o = out.cpu().numpy().astype("int64") & 0xffffffff
    hw, xcc = o[:, 0], o[:, 1] & 0xf
    cu, sh, se = (hw >> 8) & 0xf, (hw >> 12) & 1, (hw >> 13) & 7
    ids = set(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist()))
    per_xcc = [sum(1 for i in ids if i[0] == x) for x in range(8)]
    return len(ids), per_xcc


def time_plan(plan, stream, reps=3):
    best = 1e9
    with torch.cuda.stream(stream):
        plan.launch()
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
            e0.record(stream)
            plan.launch()
            e1.record(stream)
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1))
    return best            # ms for the plan's launches


# ---- the kernels: each maker returns (plan, launches in the plan, bytes per launch, flops per launch, keep-alive)
def conv_case(k, cin, cout, hw, reps, bn=False, lrelu=False, stats=False, bias=False, e_relu=False, pitch_in=None, pitch_out=None, n=16):
    pad = k // 2 if k == 3 else (1 if k == 4 else 0)
    pitch_in = pitch_in or (cin + 7) // 8 * 8
    ho = hw + 2 * pad - k + 1
    pitch_out = pitch_out or (cout + 7) // 8 * 8
    x = (torch.randn(n, hw, hw, pitch_in, device=DEV) * 0.7).to(torch.float16)
    y = torch.empty(n, ho, ho, pitch_out, dtype=torch.float16, device=DEV)
    wt = torch.randn(cout, cin, k, k, device=DEV) * (2.0 / (cin * k * k)) ** 0.5
    pw = E.PackedWeight(wt, cout, cin, k)
    pw.pack()
    b = torch.randn(cout, device=DEV) if bias else None
    keep = [x, y, wt, pw, b]
    pro = None
    if bn or lrelu:
        kw = dict(act=L.ACT_LEAKY02 if lrelu else L.ACT_RELU)
        if bn:
            keep += [torch.randn(cin, device=DEV) * 0.1, torch.rand(cin, device=DEV) + 0.5, torch.rand(cin, device=DEV) + 0.5, torch.randn(cin, device=DEV) * 0.1]
            kw.update(mean=keep[-4], var=keep[-3], gamma=keep[-2], beta=keep[-1])
        pro = E.make_prologue(**kw)
        keep.append(pro)
    ws = torch.empty(1 << 23, dtype=torch.float32, device=DEV) if stats else None
    desc = E.conv_desc(k, 1, pad, L.ACT_RELU if e_relu else L.ACT_NONE, False, cout=cout, w_layout=pw.layout)
    xv, yv = E.View(x, 0, cin), E.View(y, 0, cout)
    plan = E.Plan()
    with plan.record():
        for _ in range(reps):
            E.conv2d(xv.fd, pw, b, pro, yv.fd, desc, ws)
    keep += [ws, xv, yv, desc]
    return plan, reps, n * hw * hw * cin * 2 + n * ho * ho * cout * 2, 2.0 * n * ho * ho * cout * cin * k * k, keep, plan.kernel_names()[0]


def bwdw_case(hw, c, reps, n=16):
    pitch = (c + 127) // 128 * 128
    x = torch.randn(n, hw, hw, pitch, device=DEV).half()
    G = torch.zeros(n, hw, hw, pitch, device=DEV).bfloat16()
    dy = (torch.randn(n, hw, hw, 128, device=DEV) * 0.1).bfloat16()
    yb = torch.randn(n, hw, hw, 128, device=DEV).half()
    w = torch.randn(128, c, 1, 1, device=DEV) * 0.05
    pw = E.PackedWeight(w, c, 128, 1, transposed=False, flip=True, stride=1, layout=L.WLAYOUT_CHUNK32)
    pw.pack()
    keep = [torch.zeros(c, device=DEV), torch.ones(c, device=DEV), torch.ones(c, device=DEV), torch.zeros(c, device=DEV)]
    pro = E.make_prologue(mean=keep[0], var=keep[1], gamma=keep[2], beta=keep[3], act=1)
    ws_bn, ws = torch.empty(1 << 22, device=DEV), torch.empty(1 << 26, device=DEV)
    dw = torch.zeros(128, c, device=DEV)
    cB, cC = torch.randn(128, device=DEV) * 0.01, torch.randn(128, device=DEV) * 0.01
    xv, gv, dv, ybv = E.View(x, 0, c), E.View(G, 0, c), E.View(dy), E.View(yb)
    plan = E.Plan()
    with plan.record():
        for _ in range(reps):
            assert E.conv1x1_bwd_data_weight(dv.fd, pw, xv.fd, pro, gv.fd, ws_bn, 1, ws, dw, True, dy_affine=(ybv.fd, cB, cC)) is not None
    per = len(plan) // reps
    keep += [x, G, dy, yb, w, pw, pro, ws_bn, ws, dw, cB, cC, xv, gv, dv, ybv]
    return plan, reps, n * hw * hw * (2 * 128 + 3 * c) * 2, 2.0 * 2 * n * hw * hw * 128 * c, keep, "conv1x1_bwd_wgrad_stream(+reduce x%d)" % (per - 1)


HBM_CASES = [
    ("bwdw 256^2 C=128", lambda r: bwdw_case(256, 128, r)),
    ("ds   256^2 128->128", lambda r: conv_case(1, 128, 128, 256, r, bn=True, stats=True, pitch_in=256)),
    ("rs2  256^2 128->32 3x3", lambda r: conv_case(3, 128, 32, 256, r, bn=True, stats=True, pitch_out=256)),
]
MFMA_CASES = [
    ("D 4x4 144->288 @128^2", lambda r: conv_case(4, 144, 288, 128, r, bn=True, lrelu=True)),
    ("VGG 3x3 64->64 @256^2", lambda r: conv_case(3, 64, 64, 256, r, bias=True, e_relu=True)),
]


def main():
    print("# device:", torch.cuda.get_device_name(0), " CUs:", torch.cuda.get_device_properties(0).multi_processor_count)
    full = torch.cuda.Stream(device=DEV)
    # ---- 1. census: which mask bits are which CUs
    print("\n## 1. census: distinct (XCC, SE, SH, CU) that ran a workgroup, and how many per XCC")
    print("unmasked stream: %d CUs, per XCC %s" % census(full))
    style = None
    for st in ("linear", "perword"):
        for n in (64, 128, 192):
            got, per = census(masked_stream(bits_for(n, st)))
            print("%-8s mask of %3d bits: %3d CUs, per XCC %s" % (st, n, got, per))
            if n == 64 and style is None and got == 64 and max(per) - min(per) <= 2:
                style = st
    if style is None:
        style = "linear"
        print("!! no mask style gave an even XCC spread; using 'linear'")
    print("mask style used below: %s (n CUs evenly over the 8 XCCs)" % style)
    lo_hi = census(masked_stream(bits_for(64, style, lo=192)))
    print("%-8s mask of 64 bits at slot 192: %3d CUs, per XCC %s" % (style, lo_hi[0], lo_hi[1]))

    counts = [256, 192, 128, 64] if QUICK else [256, 224, 192, 160, 128, 96, 64, 32]
    # ---- 2. rate vs CU count
    print("\n## 2. achieved rate vs CUs given (masked stream + fdgan_set_cu_budget; best of 3 plans of `reps` launches)")
    src = torch.empty(1 << 30, dtype=torch.uint8, device=DEV)
    dst = torch.empty(1 << 30, dtype=torch.uint8, device=DEV)
    row = []
    for n in counts:
        s = masked_stream(bits_for(n, style))
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
            with torch.cuda.stream(s):
                H.cumask_copy(C.c_void_p(s.cuda_stream), C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), 1 << 30, n * 8)
                e0.record(s)
                H.cumask_copy(C.c_void_p(s.cuda_stream), C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), 1 << 30, n * 8)
                e1.record(s)
                e1.synchronize()
            best = min(best, e0.elapsed_time(e1))
        row.append(2.0 * (1 << 30) / best / 1e9)
    print("%-28s" % "CUs" + "".join("%9d" % n for n in counts))
    print("%-28s" % "copy 1 GiB (read+write) TB/s" + "".join("%9.2f" % v for v in row))
    del src, dst
    table = {}
    for name, mk in HBM_CASES + MFMA_CASES:
        vals = []
        for n in counts:
            s = masked_stream(bits_for(n, style))
            with torch.cuda.stream(s), E.cu_budget(n):
                plan, reps, byt, fl, keep, kname = mk(6 if QUICK else 10)
            ms = time_plan(plan, s)
            vals.append((ms / reps * 1e3, byt * reps / ms / 1e9, fl * reps / ms / 1e12))
            del plan, keep
        table[name] = vals
        hbm = (name, mk) in HBM_CASES
        print("%-28s" % (name + (" TB/s" if hbm else " PF/s")) + "".join("%9.2f" % (v[1] if hbm else v[2]) for v in vals) + "   [" + kname + "]")
        print("%-28s" % "   us per launch" + "".join("%9.1f" % v[0] for v in vals))

    # ---- 3. side by side
    print("\n## 3. an HBM-bound and an MFMA-bound kernel side by side (ms for the pair's work; lower is better)")
    pairs = [(HBM_CASES[0], MFMA_CASES[0]), (HBM_CASES[1], MFMA_CASES[1])] if not QUICK else [(HBM_CASES[0], MFMA_CASES[0])]
    for (hn, hmk), (mn, mmk) in pairs:
        print("pair: %s  +  %s" % (hn, mn))
        # full-chip plans: sequential and concurrent on two unmasked streams (today's behaviour)
        s1, s2 = torch.cuda.Stream(device=DEV), torch.cuda.Stream(device=DEV)
        ph, rh, _, _, kh, _ = hmk(20)
        pm, rm, _, _, km, _ = mmk(20)
        th, tm = time_plan(ph, s1), time_plan(pm, s2)
        # balance the repetitions so that both sides take about the same time alone on the full chip
        rm2 = max(1, int(round(20 * th / tm)))
        pm, rm, _, _, km, _ = mmk(rm2)
        tm = time_plan(pm, s2)

        def together(pa, sa, pb, sb):
            best = 1e9
            for _ in range(3):
                torch.cuda.synchronize()
                e0 = torch.cuda.Event(True)
                e0.record(torch.cuda.current_stream(DEV))
                sa.wait_stream(torch.cuda.current_stream(DEV))
                sb.wait_stream(torch.cuda.current_stream(DEV))
                with torch.cuda.stream(sa):
                    pa.launch()
                with torch.cuda.stream(sb):
                    pb.launch()
                cur = torch.cuda.current_stream(DEV)
                cur.wait_stream(sa)
                cur.wait_stream(sb)
                e1 = torch.cuda.Event(True)
                e1.record(cur)
                e1.synchronize()
                best = min(best, e0.elapsed_time(e1))
            return best
        print("  full chip, one after the other : %.3f + %.3f = %.3f ms" % (th, tm, th + tm))
        print("  full chip, two unmasked streams: %.3f ms" % together(ph, s1, pm, s2))
        for nh in ([192, 128] if QUICK else [224, 192, 160, 128, 96]):
            nm = 256 - nh
            sh_, sm_ = masked_stream(bits_for(nh, style)), masked_stream(bits_for(nm, style, lo=nh))
            with torch.cuda.stream(sh_), E.cu_budget(nh):
                ph2, _, _, _, kh2, _ = hmk(20)
            with torch.cuda.stream(sm_), E.cu_budget(nm):
                pm2, _, _, _, km2, _ = mmk(rm2)
            a, b = time_plan(ph2, sh_), time_plan(pm2, sm_)
            print("  %3d + %3d CUs: alone %.3f / %.3f ms, side by side %.3f ms" % (nh, nm, a, b, together(ph2, sh_, pm2, sm_)))
            del ph2, pm2, kh2, km2
        del ph, pm, kh, km


if __name__ == "__main__":
    main()
