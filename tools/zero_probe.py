"""Which gradient buffers does a backward walk still zero (bytes), per network?  (experiment aid)   python tools/zero_probe.py"""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fd-gan_amd")]
import torch
import train as T
warnings.simplefilter("ignore")
dev = torch.device("cuda:0")
ts = T.TrainStep(dev)
gt = torch.rand(16, 3, 256, 256, device=dev); haze = (gt * 0.6 + 0.3).clamp(0, 1)
for _ in range(3): ts.step(haze, gt)
torch.cuda.synchronize()
for name, m in (("netG", ts.netG), ("netD", ts.netD), ("vgg", ts.vgg)):
    for key, pl in m.__dict__.get("_plans", {}).items():
        b = getattr(pl, "_bwd", None)
        if b is None: continue
        seen, rows = set(), []
        for ptr, g in b.gbuf.items():
            if id(g) in seen: continue
            seen.add(id(g))
            rows.append((g.numel() * g.element_size() / 2**20, tuple(g.shape), id(g) in b.nozero))
        z = sum(r[0] for r in rows if not r[2]); nz = sum(r[0] for r in rows if r[2])
        print("%s plan %s: zeroed per walk %.0f MiB, not zeroed %.0f MiB" % (name, key, z, nz))
        for mb, shp, skip in sorted(rows, reverse=True)[:14]:
            print("     %7.1f MiB %-24s %s" % (mb, shp, "stored (no zeroing)" if skip else "ZEROED"))
for key, pl in ts.netD.__dict__.get("_plans", {}).items():
    b = pl._bwd
    for i, r in enumerate(b.recs):
        if r["kind"] == "conv":
            print(i, "conv", tuple(r["x"].shape), "c0", r["x"].c0, "->", None if r["y"] is None else tuple(r["y"].shape), "k", r["k"], "s", r["stride"], "sole", r.get("_sole"), "pool", r["pro"]._meta["pool"] if r.get("pro") is not None else None)
        else:
            print(i, r["kind"])
    break
