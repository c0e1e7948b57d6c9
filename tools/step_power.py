#!/usr/bin/env python
"""Board power and shader clock while the TRAINING STEP runs (is the step power-throttled?).
    python tools/step_power.py [seconds]
Polls `rocm-smi --showclocks --showpower` from a thread while TrainStep.step (B = 16 @ 256^2) loops; prints every sample."""
import os, re, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fd-gan_amd")]
import torch
import train
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
dev = torch.device("cuda:0")
ts = train.TrainStep(dev, synthetic=True)
gt = torch.rand(16, 3, 256, 256, device=dev)
haze = (gt * 0.6 + 0.3).clamp(0, 1)
for _ in range(5):
    ts.step(haze, gt)
torch.cuda.synchronize()
samples, stop = [], False
def poll():
    while not stop:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
        sclk = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", out)
        mclk = re.search(r"mclk clock level: \d+: \((\d+)Mhz\)", out)
        pw = re.search(r"Power \(W\): ([\d.]+)", out)
        samples.append((time.time(), sclk and int(sclk.group(1)), mclk and int(mclk.group(1)), pw and float(pw.group(1))))
th = threading.Thread(target=poll); th.start()
t0, n = time.time(), 0
while time.time() - t0 < seconds:
    for _ in range(10):
        ts.step(haze, gt, sync=False)
    torch.cuda.synchronize()
    n += 10
t1 = time.time()
stop = True; th.join()
print("steps", n, "mean ms/step", round((t1 - t0) / n * 1e3, 3))
for t, s, m, p in samples:
    print("  t=%.2fs sclk %s MHz mclk %s MHz power %s W" % (t - t0, s, m, p))
ok = [p for _, _, _, p in samples if p]
sc = [s for _, s, _, _ in samples if s]
if ok:
    print("power W: min %.0f mean %.0f max %.0f | sclk MHz: min %d mean %d max %d" % (min(ok), sum(ok) / len(ok), max(ok), min(sc), sum(sc) / len(sc), max(sc)))
