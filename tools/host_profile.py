"""Where does the HOST spend its time enqueueing one training step?  cProfile over a few un-synchronised steps (experiment aid)."""
import cProfile, os, pstats, sys, io
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [R, os.path.join(R, "fd-gan_amd")]
import numpy as np, torch
import train as train_mod
from fdgan_hip import dp as dpm
dev = torch.device("cuda:0")
ts = train_mod.TrainStep(dev, dp=None, synthetic=True)
gt = torch.from_numpy(np.random.default_rng(99).random((16, 3, 256, 256), dtype=np.float32)).to(dev)
haze = (gt * 0.6 + 0.3).clamp(0, 1)
for _ in range(4):
    ts.step(haze, gt)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    ts.step(haze, gt, sync=False)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
st = pstats.Stats(pr, stream=s)
st.sort_stats("tottime").print_stats(28)
st.sort_stats("cumulative").print_stats(40)
print(s.getvalue())
