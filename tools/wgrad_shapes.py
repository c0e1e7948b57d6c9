"""Per-shape time of the weight-gradient launches of one training step (experiment aid): wraps
engine.conv_bwd_weight with stream events and prints the table sorted by total time."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fd-gan_amd")]
import torch
from fdgan_hip import engine as E
import train as T

rec = []
orig = E.conv_bwd_weight
def timed(x_fd, pro, dy_fd, desc, dw, dbias=None, ws=None, accumulate=False):
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    orig(x_fd, pro, dy_fd, desc, dw, dbias, ws, accumulate)
    e1.record()
    pool = bool(pro.pool2) if pro is not None else False
    rec.append(((int(x_fd.n), int(dy_fd.h), int(dy_fd.w), int(x_fd.c), int(dy_fd.c), int(desc.ksize), int(desc.stride), pool,
                 ws is not None), e0, e1))
E.conv_bwd_weight = timed
orig_job = E.conv_bwd_weight_job
def timed_job(x_fd, pro, dy_fd, desc, dw, ws, defer, accumulate=False):
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    r = orig_job(x_fd, pro, dy_fd, desc, dw, ws, defer, accumulate)
    e1.record()
    rec.append(((int(x_fd.n), int(dy_fd.h), int(dy_fd.w), int(x_fd.c), int(dy_fd.c), int(desc.ksize), int(desc.stride), False, True), e0, e1))
    return r
E.conv_bwd_weight_job = timed_job
os.environ["FDGAN_NO_WGRAD_STREAM"] = "1"      # events of a launch on the walk's own stream
import fdgan_hip.backward as BW
BW.E = E
BW.FORCE_EAGER = True                          # a recorded walk replays launches; the timed wrappers above need the eager walk

B, S = 16, 256
ts = T.TrainStep(device="cuda:0") if "device" in T.TrainStep.__init__.__code__.co_varnames else T.TrainStep()
haze = torch.rand(B, 3, S, S, device="cuda:0"); gt = torch.rand(B, 3, S, S, device="cuda:0")
for _ in range(2): ts.step(haze, gt)
rec.clear()
ts.step(haze, gt)
torch.cuda.synchronize()
agg = collections.OrderedDict()
for key, e0, e1 in rec:
    a = agg.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += e0.elapsed_time(e1) * 1e3
tot = sum(v[1] for v in agg.values())
print(f"total {tot/1e3:.2f} ms in {len(rec)} launches")
print("   N   Ho   Wo   Cin Cout k s pool ws | calls  total_us  avg_us  TFLOP/s")
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    N, Ho, Wo, Cin, Cout, ks, st, pool, ws = k
    fl = 2.0 * N * Ho * Wo * Cin * Cout * ks * ks
    print(f"{N:4d} {Ho:4d} {Wo:4d} {Cin:5d} {Cout:4d} {ks} {st} {int(pool)}    {int(ws)}  | {n:4d} {t:9.0f} {t/n:8.1f} {fl*n/t/1e6:7.1f}")
