"""Experiment: how fast is the training step when the host is taken out of it?  Captures one whole TrainStep.step() into a HIP graph
(torch.cuda.graph: every kernel of both streams, with their cross-stream waits) and replays it.  NOT a valid training loop as it
stands -- the image pool's host-side coin flips and Adam's step count are frozen into the graph -- only a measurement of the
GPU-side time of a step that needs no Python between its launches."""
import os, sys, time, faulthandler; faulthandler.enable()
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [R, os.path.join(R, "fd-gan_amd")]
import numpy as np, torch
import train as train_mod
dev = torch.device("cuda:0")
ts = train_mod.TrainStep(dev, dp=None, synthetic=True)
gt = torch.from_numpy(np.random.default_rng(99).random((16, 3, 256, 256), dtype=np.float32)).to(dev)
haze = (gt * 0.6 + 0.3).clamp(0, 1)
for _ in range(6):
    ts.step(haze, gt)
torch.cuda.synchronize()
def timed(fn, n=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("eager            %.3f ms/step" % timed(lambda: ts.step(haze, gt, sync=False)), flush=True)
ts.pool.query = lambda x: x                     # host-side randomness out (measurement only)
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream(device=dev)
s.wait_stream(torch.cuda.current_stream(dev))
with torch.cuda.stream(s):
    for _ in range(3):
        ts.step(haze, gt, sync=False)
torch.cuda.current_stream(dev).wait_stream(s)
torch.cuda.synchronize()
try:
    with torch.cuda.graph(g, stream=s):
        out = ts.step(haze, gt, sync=False)
except Exception as e:
    print("capture failed:", type(e).__name__, str(e)[:600])
    raise SystemExit(1)
torch.cuda.synchronize()
print("graph replay     %.3f ms/step" % timed(g.replay))
print("losses", dict(zip(ts.LOSS_NAMES, [round(v, 4) for v in out.tolist()])))
