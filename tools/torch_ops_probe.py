"""Which torch operators still launch GPU work inside a training step (everything else is library launches)?  (experiment aid)
    python tools/torch_ops_probe.py
Round 4: about 40 per step (12 mul, 6 copy_, 5 fill_, 5 add_, 4 add, 2 cat, sum / div / sub / neg / mul_): the loss arithmetic and its
gradient scaling, ~0.1 ms."""
import os, sys, warnings, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fd-gan_amd")]
import torch
import train as T
warnings.simplefilter("ignore")
dev = torch.device("cuda:0")
ts = T.TrainStep(dev)
gt = torch.rand(16, 3, 256, 256, device=dev); haze = (gt * 0.6 + 0.3).clamp(0, 1)
for _ in range(6): ts.step(haze, gt)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    ts.step(haze, gt)
    torch.cuda.synchronize()
agg = collections.Counter()
stacks = collections.defaultdict(collections.Counter)
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CPU and ev.name.startswith("aten::") and len(ev.kernels) > 0:
        agg[ev.name] += 1
        st = [s for s in (ev.stack or []) if "fd-gan_amd" in s or "bench.py" in s]
        stacks[ev.name][st[0] if st else "?"] += 1
for name, n in agg.most_common(12):
    print("%-28s %4d" % (name, n))
    for s, c in stacks[name].most_common(6):
        print("      %3d  %s" % (c, s[-110:]))
