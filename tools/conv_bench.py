#!/usr/bin/env python
"""Micro-benchmark of one fused conv launch through the C ABI (kernel tuning aid).

  python tools/conv_bench.py --k 3 --cin 128 --cout 32 --n 16 --hw 256 --bn --stats --pitch-out 256
Prints one JSON line per configuration: average launch time (hipEvents around each launch
of a recorded plan), algorithmic GB/s and TFLOP/s.  `--suite netg` runs the netG hot shapes.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fd-gan_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
from fdgan_hip import engine as E  # noqa: E402
from fdgan_hip import lib as L  # noqa: E402


def run(k, cin, cout, n, h, w, bn=False, relu=False, stats=False, pool=False, pitch_in=None, pitch_out=None,
        bias=False, layout=None, reps=20, upsample=False, lrelu=False, e_relu=False, keep_out=False):
    dev = torch.device("cuda:0")
    pad = k // 2 if k == 3 else (1 if k == 4 else 0)
    pitch_in = pitch_in or (cin + 7) // 8 * 8
    ho, wo = (h // 2, w // 2) if pool else (h + 2 * pad - k + 1, w + 2 * pad - k + 1)
    up = 2 if upsample else 1
    pitch_out = pitch_out or (cout + 7) // 8 * 8
    torch.manual_seed(1234)
    x = (torch.randn(n, h, w, pitch_in, device=dev) * 0.7).to(torch.float16)
    y = torch.empty(n, ho * up, wo * up, pitch_out, dtype=torch.float16, device=dev)
    wt = torch.randn(cout, cin, k, k, device=dev) * (2.0 / (cin * k * k)) ** 0.5
    pw = E.PackedWeight(wt, cout, cin, k, layout=layout)
    pw.pack()
    b = torch.randn(cout, device=dev) if bias else None
    pro = None
    keep = []
    if bn or relu or pool or lrelu:
        kw = dict(act=L.ACT_LEAKY02 if lrelu else (L.ACT_RELU if (relu or bn) else L.ACT_NONE), pool=pool)
        if bn:
            mean, var = torch.randn(cin, device=dev) * 0.1, torch.rand(cin, device=dev) + 0.5
            g, bt = torch.rand(cin, device=dev) + 0.5, torch.randn(cin, device=dev) * 0.1
            keep += [mean, var, g, bt]
            kw.update(mean=mean, var=var, gamma=g, beta=bt)
        pro = E.make_prologue(**kw)
    ws = torch.empty(1 << 23, dtype=torch.float32, device=dev) if stats else None
    desc = E.conv_desc(k, 1, pad, L.ACT_RELU if e_relu else L.ACT_NONE, upsample, cout=cout, w_layout=pw.layout)
    xv, yv = E.View(x, 0, cin), E.View(y, 0, (cout + 3) // 4 * 4 if (cout + 3) // 4 * 4 <= pitch_out else cout)
    plan = E.Plan()
    with plan.record():
        for _ in range(reps):
            E.conv2d(xv.fd, pw, b, pro, yv.fd, desc, ws)
    dbg = None
    if os.environ.get("FDGAN_TIMING"):
        dbg = torch.zeros(64, dtype=torch.int64, device=dev)
        L.load().fdgan_debug_timing(dbg.data_ptr())
        E.conv2d(xv.fd, pw, b, pro, yv.fd, desc, ws)
        torch.cuda.synchronize()
        L.load().fdgan_debug_timing(None)
        print("phase cycles per wave [mfma, staging, barrier, epilogue, steps]:", dbg.view(8, 8)[:, :6].tolist())
    for _ in range(2):
        plan.launch()
    torch.cuda.synchronize()
    ms = plan.profile()
    ms = sorted(ms)[len(ms) // 4: -len(ms) // 4 or None]          # inter-quartile mean
    t = sum(ms) / len(ms) * 1e-3
    byt = n * h * w * cin * 2 + n * ho * up * wo * up * cout * 2
    fl = 2.0 * n * ho * wo * (4 if pool else 1) * cout * cin * k * k
    res = {"kernel": plan.kernel_names()[0], "shape": "%d->%d k%d @%dx%d n%d" % (cin, cout, k, h, w, n),
           "us": round(t * 1e6, 2), "GB/s": round(byt / t / 1e9, 1), "TFLOP/s": round(fl / t / 1e12, 1),
           "MB": round(byt / 1e6, 1)}
    if keep_out:
        res["_y"] = y
        res["_stats"] = ws[:4096].clone() if ws is not None else None
    return res


# MFMA-bound shapes of the training step (VGG16, Fusion-D, the generator's wide 3x3 convs), batch 16
MFMA_SHAPES = [
    dict(k=3, cin=160, cout=128, n=16, h=128, w=128, bias=True, stats=True, pitch_out=512),     # conv_refine4
    dict(k=3, cin=640, cout=512, n=16, h=32, w=32, bias=True, pitch_out=768),                   # conv_refin6
    dict(k=3, cin=1024, cout=256, n=16, h=32, w=32, relu=True, pitch_out=768),                  # dense_block4.conv2
    dict(k=3, cin=512, cout=128, n=16, h=64, w=64, relu=True, pitch_out=512),                   # dense_block5.conv2
    dict(k=3, cin=64, cout=64, n=16, h=256, w=256, bias=True, e_relu=True),                     # VGG conv1_2
    dict(k=3, cin=64, cout=128, n=16, h=128, w=128, bias=True, e_relu=True),                    # VGG conv2_1
    dict(k=3, cin=128, cout=128, n=16, h=128, w=128, bias=True, e_relu=True),                   # VGG conv2_2
    dict(k=3, cin=128, cout=256, n=16, h=64, w=64, bias=True, e_relu=True),                     # VGG conv3_1
    dict(k=3, cin=256, cout=256, n=16, h=64, w=64, bias=True, e_relu=True),                     # VGG conv3_2/3
    dict(k=3, cin=256, cout=512, n=16, h=32, w=32, bias=True, e_relu=True),                     # VGG conv4_1
    dict(k=3, cin=512, cout=512, n=16, h=32, w=32, bias=True, e_relu=True),                     # VGG conv4_2/3
    dict(k=3, cin=36, cout=72, n=16, h=128, w=128, lrelu=True, stats=True, pitch_in=40),         # D layer2
    dict(k=3, cin=72, cout=144, n=16, h=128, w=128, bn=True, lrelu=True, stats=True),           # D layer3
    dict(k=4, cin=144, cout=288, n=16, h=128, w=128, bn=True, lrelu=True),                      # D layer4
]


def ab_suite(shapes, var="FDGAN_DEBUG_WD", variants=("",), rounds=3):
    """A/B of a tuning switch (needs a FDGAN_TUNING=1 build): per shape old (switch = 0) vs each variant, interleaved
    over `rounds` rounds (the clock state drifts by several % between back-to-back measurements: best-of and median are
    reported), outputs compared."""
    import statistics
    for cfg in shapes:
        names, times, outs = {}, {}, {}
        order = ["0"] + list(variants)
        for r in range(rounds):
            for v in order:
                if v:
                    os.environ[var] = v
                else:
                    os.environ.pop(var, None)
                res = run(keep_out=(r == 0), **cfg)
                if r == 0:
                    outs[v] = res.pop("_y").float()[..., :cfg["cout"]]
                    res.pop("_stats", None)
                    names[v] = res["kernel"]
                times.setdefault(v, []).append(res["us"])
                flop_us = res["TFLOP/s"] * res["us"]
        os.environ.pop(var, None)
        row = {"shape": res["shape"]}
        for v in order:
            best, med = min(times[v]), statistics.median(times[v])
            rel = float((outs["0"] - outs[v]).norm() / (outs["0"].norm() + 1e-30))
            row[v or "new"] = "%s best %.1fus %.0fTF med %.1fus x%.2f d%.0e" % (names[v], best, flop_us / best, med,
                                                                               min(times["0"]) / best, rel)
        print(json.dumps(row), flush=True)


HEADLINE_SHAPES = [
    dict(k=3, cin=128, cout=32, n=16, h=256, w=256, bn=True, stats=True, pitch_out=256),
    dict(k=3, cin=128, cout=32, n=16, h=128, w=128, bn=True, stats=True, pitch_out=512),
    dict(k=3, cin=128, cout=32, n=16, h=64, w=64, bn=True, stats=True, pitch_out=1024),
]

SUITES = {
    "netg": [
        dict(k=3, cin=128, cout=32, n=16, h=256, w=256, bn=True, stats=True, pitch_out=256),
        dict(k=3, cin=128, cout=32, n=16, h=128, w=128, bn=True, stats=True, pitch_out=512),
        dict(k=3, cin=128, cout=32, n=16, h=64, w=64, bn=True, stats=True, pitch_out=1024),
        dict(k=1, cin=64, cout=128, n=16, h=256, w=256, bn=True, stats=True, pitch_in=256),
        dict(k=1, cin=128, cout=128, n=16, h=256, w=256, bn=True, stats=True, pitch_in=256),
        dict(k=1, cin=224, cout=128, n=16, h=256, w=256, bn=True, stats=True, pitch_in=256),
        dict(k=1, cin=480, cout=128, n=16, h=128, w=128, bn=True, stats=True, pitch_in=512),
        dict(k=1, cin=992, cout=128, n=16, h=64, w=64, bn=True, stats=True, pitch_in=1024),
        dict(k=1, cin=256, cout=128, n=16, h=256, w=256, bn=True, pool=True, pitch_out=160),
        dict(k=3, cin=160, cout=128, n=16, h=128, w=128, bias=True, stats=True, pitch_out=512),
        dict(k=3, cin=640, cout=512, n=16, h=32, w=32, bias=True, pitch_out=768),
        dict(k=3, cin=1024, cout=256, n=16, h=32, w=32, relu=True, pitch_out=768),
    ],
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--suite", default="")
    for nm in ("k", "cin", "cout", "n", "hw", "pitch-in", "pitch-out", "reps"):
        ap.add_argument("--" + nm, type=int, default=None)
    for nm in ("bn", "relu", "stats", "pool", "bias"):
        ap.add_argument("--" + nm, action="store_true")
    ap.add_argument("--layout", type=int, default=None)
    ap.add_argument("--variants", default="")
    a = ap.parse_args()
    L.load()
    if a.suite == "headline_ab":
        ab_suite(HEADLINE_SHAPES, variants=tuple(a.variants.split(",")))
        return
    if a.suite == "mfma_ab":
        ab_suite(MFMA_SHAPES, variants=tuple(a.variants.split(",")))
        return
    if a.suite == "mfma":      # the MFMA-bound shapes, `--reps` launches back to back each (sustained clocks: a 20-launch run is still ramping up)
        for cfg in MFMA_SHAPES:
            print(json.dumps(run(reps=a.reps or 300, **cfg)), flush=True)
        return
    if a.suite:
        for cfg in SUITES[a.suite]:
            print(json.dumps(run(**cfg)), flush=True)
        return
    print(json.dumps(run(a.k, a.cin, a.cout, a.n or 16, a.hw, a.hw, a.bn, a.relu, a.stats, a.pool, a.pitch_in,
                         a.pitch_out, a.bias, a.layout, a.reps or 20)))


if __name__ == "__main__":
    main()
