"""Upper bound on what a faster weight-gradient kernel can buy (experiment aid; the step's results are WRONG while a group is
skipped): drops the weight-gradient launches of one group of shapes and times the full training step.
    python tools/skip_wgrad_probe.py           # baseline + every group
Groups: D (Fusion-discriminator convs), edge (generator convs outside the dense blocks, 3x3), pool1x1 (transition 1x1 with pooled
input), other1x1, dense3x3 (growth convs)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fd-gan_amd")]
import torch
from fdgan_hip import engine as E
import fdgan_hip.backward as BW
import train as T

D_SHAPES = {(9, 40, 4), (36, 72, 3), (72, 144, 3), (144, 288, 4), (288, 1, 4)}
def group(x_fd, pro, dy_fd, desc):
    cin, cout, k = int(x_fd.c), int(dy_fd.c), int(desc.ksize)
    if (cin, cout, k) in D_SHAPES: return "D"
    if k == 3 and cin == 128 and cout == 32: return "dense3x3"
    if k == 3: return "edge"
    if k == 1 and pro is not None and bool(pro.pool2): return "pool1x1"
    return "other1x1"

SKIP = set()
seen = {}
o1, o2 = E.conv_bwd_weight, E.conv_bwd_weight_job
def w1(x_fd, pro, dy_fd, desc, dw, dbias=None, ws=None, accumulate=False):
    g = group(x_fd, pro, dy_fd, desc); seen[g] = seen.get(g, 0) + 1
    if g in SKIP: return
    return o1(x_fd, pro, dy_fd, desc, dw, dbias, ws, accumulate)
def w2(x_fd, pro, dy_fd, desc, dw, ws, defer, accumulate=False):
    g = group(x_fd, pro, dy_fd, desc); seen[g] = seen.get(g, 0) + 1
    if g in SKIP: return None
    return o2(x_fd, pro, dy_fd, desc, dw, ws, defer, accumulate)
E.conv_bwd_weight, E.conv_bwd_weight_job = w1, w2

def measure(skip):
    SKIP.clear(); SKIP.update(skip); seen.clear()
    torch.manual_seed(0)
    ts = T.TrainStep("cuda:0")
    haze = torch.rand(16, 3, 256, 256, device="cuda:0"); gt = torch.rand(16, 3, 256, 256, device="cuda:0")
    for _ in range(5): ts.step(haze, gt)          # two eager walks, the recording, two replays
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 20
    for _ in range(n): ts.step(haze, gt)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

import warnings; warnings.simplefilter("ignore")
base = measure(())
print("baseline %.2f ms   launches per group in the first steps: %s" % (base, dict(seen)))
for g in (("D",), ("edge",), ("pool1x1",), ("other1x1",), ("dense3x3",), ("D", "edge", "pool1x1", "other1x1")):
    t = measure(g)
    print("without %-34s %.2f ms  (%+.2f)" % ("+".join(g), t, t - base))
