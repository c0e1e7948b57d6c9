"""Forward time of the legacy DCPDN networks on the HIP path and on the CPU oracle:
python tools/legacy_bench.py [G|G2|Dense|dehaze] [nf] [batch]   -> one JSON line"""
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
net = (net22.Dense() if kind == "Dense" else getattr(net22, kind)(3, 3, nf)).cuda()
x = torch.rand(B, 3, 256, 256, device="cuda")
res = {"net": "dehaze22.%s" % kind + ("()" if kind == "Dense" else "(3, 3, %d)" % nf), "batch": B, "image": [3, 256, 256]}
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
    plans = [net.hip_plan(x)] if kind != "dehaze" else [net.hip_plan(x), net.tran_dense.hip_plan(x), net.atp_est.hip_plan(x)]
    gf = sum(m["flops"] for P in plans for m in P.meta) / 1e9
    res[mode] = {"ms": round(ms, 3), "images_per_s": round(B / ms * 1e3, 1), "gflop_reference_formulation": round(gf, 1),
                 "tflops": round(gf / ms, 1), "launches": sum(len(P.main) for P in plans)}
# CPU oracle, batch 2 (eval): the same network in fp32 torch on the host
from oracle import legacy_ref
sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
xc = torch.rand(2, 3, 256, 256)
torch.set_num_threads(16)
cpu = {"G": lambda: legacy_ref.unet_forward(sd, xc, False, "G"), "G2": lambda: legacy_ref.unet_forward(sd, xc, False, "G2"),
       "Dense": lambda: legacy_ref.dense_forward(sd, xc, False, "pyramid"), "dehaze": lambda: legacy_ref.dehaze_forward(sd, xc, False)}[kind]
with torch.no_grad():
    cpu()
    t0 = time.perf_counter()
    for _ in range(3):
        cpu()
res["cpu_oracle_eval"] = {"images_per_s": round(2 * 3 / (time.perf_counter() - t0), 2), "threads": 16, "batch": 2}
print(json.dumps(res))
