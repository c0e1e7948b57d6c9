"""Forward time of the legacy U-Nets on the HIP path and on the CPU oracle:
python tools/legacy_bench.py [G|G2] [nf] [batch]   -> one JSON line"""
import json, os, sys, time, warnings
import torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "fd-gan_amd"))
sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")
import models.dehaze22 as net22
kind = sys.argv[1] if len(sys.argv) > 1 else "G"
nf = int(sys.argv[2]) if len(sys.argv) > 2 else 64
B = int(sys.argv[3]) if len(sys.argv) > 3 else 16
net = getattr(net22, kind)(3, 3, nf).cuda()
x = torch.rand(B, 3, 256, 256, device="cuda")
res = {"net": "dehaze22.%s(3, 3, %d)" % (kind, nf), "batch": B, "image": [3, 256, 256]}
for mode in ("eval", "train"):
    net.train(mode == "train")
    with torch.no_grad():
        for _ in range(3):
            net(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            net(x)
        torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    P = net.hip_plan(x)
    gf = sum(m["flops"] for m in P.meta) / 1e9
    res[mode] = {"ms": round(ms, 3), "images_per_s": round(B / ms * 1e3, 1), "gflop_reference_formulation": round(gf, 1),
                 "tflops": round(gf / ms, 1), "launches": len(P.main)}
# CPU oracle, batch 2 (eval): the same network in fp32 torch on the host
from oracle import legacy_ref
sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
xc = torch.rand(2, 3, 256, 256)
torch.set_num_threads(16)
with torch.no_grad():
    legacy_ref.unet_forward(sd, xc, False, kind)
    t0 = time.perf_counter()
    for _ in range(3):
        legacy_ref.unet_forward(sd, xc, False, kind)
res["cpu_oracle_eval"] = {"images_per_s": round(2 * 3 / (time.perf_counter() - t0), 2), "threads": 16, "batch": 2}
print(json.dumps(res))
