#!/usr/bin/env python
"""Clock / power while one conv launch is repeated for a few seconds (tuning aid: is a kernel power-limited?).

  python tools/power_probe.py [--seconds 4] -- <tools/conv_bench.py arguments of ONE configuration>
Polls `rocm-smi --showclocks --showpower` while the plan replays; prints sclk / power samples and the mean launch time."""
import os
import re
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    argv = sys.argv[1:]
    seconds = 4.0
    if argv and argv[0] == "--seconds":
        seconds = float(argv[1])
        argv = argv[2:]
    if argv and argv[0] == "--":
        argv = argv[1:]
    import argparse
    import conv_bench as cb
    import torch
    from fdgan_hip import engine as E
    ap = argparse.ArgumentParser()
    for k in ("k", "cin", "cout", "n", "hw"):
        ap.add_argument("--" + k, type=int, required=True)
    ap.add_argument("--bn", action="store_true")
    ap.add_argument("--stats", action="store_true")
    ap.add_argument("--pitch-out", type=int, default=None)
    a = ap.parse_args(argv)
    # record one plan of 200 launches through conv_bench's own set-up
    plans = []
    orig = E.Plan

    class Keep(orig):
        def __init__(self, *x, **kw):
            super().__init__(*x, **kw)
            plans.append(self)
    E.Plan = Keep
    res = cb.run(a.k, a.cin, a.cout, a.n, a.hw, a.hw, bn=a.bn, stats=a.stats, pitch_out=a.pitch_out, reps=200, keep_out=True)
    E.Plan = orig
    plan = plans[-1]
    samples, stop = [], False

    def poll():
        while not stop:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
            sclk = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", out)
            pw = re.search(r"Power \(W\): ([\d.]+)", out)
            samples.append((time.time(), int(sclk.group(1)) if sclk else None, float(pw.group(1)) if pw else None))
    th = threading.Thread(target=poll)
    th.start()
    t0 = time.time()
    nl = 0
    while time.time() - t0 < seconds:
        plan.launch()
        nl += 200
        if nl % 2000 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    t1 = time.time()
    stop = True
    th.join()
    print("launches", nl, "mean us", round((t1 - t0) / nl * 1e6, 2), "| single-plan profile us", res["us"])
    for t, s, p in samples:
        print("  t=%.2fs sclk %s MHz power %s W" % (t - t0, s, p))


if __name__ == "__main__":
    main()
