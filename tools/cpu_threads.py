import sys, time, os; sys.path.insert(0, os.getcwd())
import torch
from oracle import dehaze1113_ref as ref
from oracle.detweights import det_input, fill_state_dict
print("cpu_count", os.cpu_count(), flush=True)
og = ref.FDGAN(); fill_state_dict(og)
x = det_input((1,3,256,256))
for th in (8, 16, 32, 64, 128):
    torch.set_num_threads(th)
    with torch.no_grad():
        og(x.clone()); t=time.perf_counter(); og(x.clone()); og(x.clone()); dt=(time.perf_counter()-t)/2
    print(th, "threads: %.3f s/img" % dt, flush=True)
