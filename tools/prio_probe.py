"""Does running the step's main chain on a HIGH-priority stream (the weight-gradient / side streams stay at normal priority) shorten
the step?  (experiment aid)    python tools/prio_probe.py
Measured, round 4: no -- 27.4 / 27.6 ms at normal priority, 34.3 / 32.4 ms with the main chain on a high-priority stream (HIP range:
least 1, greatest -1; torch offers 0 and -1)."""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fd-gan_amd")]
import torch
import train as T
warnings.simplefilter("ignore")
dev = torch.device("cuda:0")
def measure(high):
    torch.manual_seed(0)
    st = torch.cuda.Stream(device=dev, priority=-1) if high else torch.cuda.current_stream(dev)
    with torch.cuda.stream(st):
        ts = T.TrainStep(dev)
        gt = torch.rand(16, 3, 256, 256, device=dev); haze = (gt * 0.6 + 0.3).clamp(0, 1)
        for _ in range(5): ts.step(haze, gt)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 30
        for _ in range(n): ts.step(haze, gt)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
for high in (False, True, False, True):
    print("main chain on a %s-priority stream: %.2f ms per step" % ("HIGH" if high else "normal", measure(high)))
