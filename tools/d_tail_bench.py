#!/usr/bin/env python
"""Times the Fusion-discriminator's tail on its own: the one-filter 4x4 conv 288 -> 1 at 127 x 127 (B = 16), forward, data gradient
and weight gradient (csrc/conv_c1.hip), `reps` launches back to back in a recorded plan (hipEvents around each launch).
    python tools/d_tail_bench.py [reps]"""
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


def timed(fn, reps):
    plan = E.Plan()
    with plan.record():
        for _ in range(reps):
            fn()
    for _ in range(2):
        plan.launch()
    torch.cuda.synchronize()
    ms = sorted(plan.profile())
    ms = ms[len(ms) // 4: -len(ms) // 4 or None]
    return sum(ms) / len(ms) * 1e3, plan.kernel_names()[0]


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    dev = torch.device("cuda:0")
    L.load()
    n, h, w, c = 16, 127, 127, 288
    torch.manual_seed(0)
    x = (torch.randn(n, h, w, c, device=dev) * 0.7).to(torch.float16)
    wt = torch.randn(1, c, 4, 4, device=dev) * 0.02
    pw = E.PackedWeight(wt, 1, c, 4)
    pw.pack()
    pro = E.make_prologue(act=L.ACT_LEAKY02)
    y = torch.empty(n, 1, h - 1, w - 1, dtype=torch.float32, device=dev)
    desc = E.conv_desc(4, 1, 1, L.ACT_NONE, False, cout=1, w_layout=pw.layout)
    xv = E.View(x, 0, c)
    us, name = timed(lambda: E.conv2d(xv.fd, pw, None, pro, E.nchw_f32_view(y), desc, None), reps)
    mb = x.numel() * 2 / 1e6
    print(json.dumps({"op": "forward 288 -> 1 4x4 @127^2 n16", "kernel": name, "us": round(us, 1), "input_MB": round(mb, 1), "TB/s": round(mb / us, 2)}))


def first_layers(reps):
    """The image-reading first layers (csrc/conv_sc.hip): 3 -> 64 3x3 @256^2 and 9 -> 36 4x4 stride 2 @256^2, B = 16."""
    dev = torch.device("cuda:0")
    for (cin, pitch, cout, k, st, hw, name) in ((3, 8, 64, 3, 1, 256, "3 -> 64 3x3 @256^2"), (9, 16, 36, 4, 2, 256, "9 -> 36 4x4 s2 @256^2")):
        x = (torch.randn(16, hw, hw, pitch, device=dev) * 0.7).to(torch.float16)
        x[..., cin:] = 0
        wt = torch.randn(cout, cin, k, k, device=dev) * 0.1
        pw = E.PackedWeight(wt, cout, cin, k, stride=st)
        pw.pack()
        ho = (hw + 2 - k) // st + 1
        cst = (cout + 7) // 8 * 8
        y = torch.empty(16, ho, ho, cst, dtype=torch.float16, device=dev)
        desc = E.conv_desc(k, st, 1, L.ACT_RELU, False, cout=cout, w_layout=pw.layout)
        b = torch.randn(cout, device=dev)
        xv, yv = E.View(x, 0, cin), E.View(y, 0, (cout + 3) // 4 * 4)
        us, kn = timed(lambda: E.conv2d(xv.fd, pw, b, None, yv.fd, desc, None), reps)
        mb = (x.numel() + y.numel()) * 2 / 1e6
        print(json.dumps({"op": name + " n16", "kernel": kn, "us": round(us, 1), "MB": round(mb, 1), "TB/s": round(mb / us, 2)}))


if __name__ == "__main__":
    main()
    first_layers(int(sys.argv[1]) if len(sys.argv) > 1 else 100)
