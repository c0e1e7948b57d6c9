"""Thread count of the CPU baseline legs (bench.py cpu_baseline / cpu_baseline_train) on the GPU box's host (experiment aid):
    python tools/cpu_threads.py            netG forward, batch 1 @256^2, 8 .. 128 threads
    python tools/cpu_threads.py train16    ONE full training step of the oracle at batch 16 @256^2 per thread count (VERDICT r5 #9:
                                           the batch-16 leg reused the batch-1 optimum)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle.detweights import det_input, fill_state_dict
print("cpu_count", os.cpu_count(), flush=True)
if len(sys.argv) > 1 and sys.argv[1] == "train16":
    from oracle.train_ref import TrainStepRef
    torch.manual_seed(0)
    ts = TrainStepRef()
    gt = det_input((16, 3, 256, 256), seed=98)
    haze = (gt * 0.6 + 0.3).clamp(0, 1)
    gt1 = det_input((1, 3, 256, 256), seed=99)
    torch.set_num_threads(16)
    ts.step((gt1 * 0.6 + 0.3).clamp(0, 1), gt1)            # warm-up
    for th in [int(a) for a in sys.argv[2:]] or (16, 32, 48, 64, 96):
        torch.set_num_threads(th)
        t = time.perf_counter()
        ts.step(haze, gt)
        dt = time.perf_counter() - t
        print("%3d threads: %.1f s per batch-16 step = %.3f images/s" % (th, dt, 16 / dt), flush=True)
    sys.exit(0)
from oracle import dehaze1113_ref as ref
og = ref.FDGAN(); fill_state_dict(og)
x = det_input((1, 3, 256, 256))
for th in (8, 16, 32, 64, 128):
    torch.set_num_threads(th)
    with torch.no_grad():
        og(x.clone()); t = time.perf_counter(); og(x.clone()); og(x.clone()); dt = (time.perf_counter() - t) / 2
    print(th, "threads: %.3f s/img" % dt, flush=True)
