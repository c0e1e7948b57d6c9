"""Debug aid: per-parameter gradient error of a legacy network's HIP backward vs the functional oracle, in reverse network order.
    python tools/dbg_legacy_bwd.py dense1113|dense2_1113|dense22|G|G2|dehaze"""
import importlib, os, sys, torch
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [R, os.path.join(R, "fd-gan_amd"), os.path.join(R, "tests")]
from oracle import legacy_ref
from oracle.detweights import det_input, fill_state_dict
which = sys.argv[1]
DEV = "cuda:0"
torch.manual_seed(0)
if which.startswith("dense"):
    mod, cls, tail = {"dense1113": ("dehaze1113", "Dense", "bn"), "dense2_1113": ("dehaze1113", "Dense2", "pyramid"), "dense22": ("dehaze22", "Dense", "pyramid")}[which]
    net = getattr(importlib.import_module("models." + mod), cls)()
    fill_state_dict(net, seed=6)
    if os.environ.get("SHIFT"):
        from oracle.detweights import shift_bn_bias
        shift_bn_bias(net, float(os.environ["SHIFT"]))
    with torch.no_grad():
        net.refine3.weight.mul_(0.1), net.refine3.bias.mul_(0.1)
    x = det_input((int(os.environ.get("N", "2")), 3, int(os.environ.get("H", "64")), int(os.environ.get("W", "96"))), seed=33)
    TAPS = {}
    fwd = lambda sdg: legacy_ref.dense_forward(sdg, x.clone(), True, tail, taps=TAPS)
    masks = None
elif which in ("G", "G2"):
    import models.dehaze22 as net22
    net = getattr(net22, which)(3, 3, 8)
    fill_state_dict(net, seed=5)
    x = det_input((2, 3, 256, 256), seed=21)
    masks = [(torch.rand(2, 64) > 0.5).float() * 2.0 for _ in range(3)]
    fwd = lambda sdg: legacy_ref.unet_forward(sdg, x.clone(), True, which, masks=list(masks))[0]
sd = {k: v.clone() for k, v in net.state_dict().items()}
cot = det_input(tuple(x.shape), seed=7, lo=-1.0, hi=1.0)
sdg = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running_" not in k else v.clone()) for k, v in sd.items()}
yo = fwd(sdg)
(yo * cot).sum().backward()
emu = None
if os.environ.get("EMU"):
    from hiputil import emulated_functional_convs
    emu = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running_" not in k else v.clone()) for k, v in sd.items()}
    with emulated_functional_convs(legacy_ref):
        (fwd(emu) * cot).sum().backward()
net = net.to(DEV).train()
if masks is not None:
    net.__dict__["_forced_dropout_masks"] = [m.to(DEV) for m in masks]
y = net(x.to(DEV))
(y * cot.to(DEV)).sum().backward()
torch.cuda.synchronize()
print("forward rel-rms vs oracle %.5f" % float((y.detach().cpu() - yo.detach()).norm() / yo.detach().norm()))
rows = []
for k, p in net.named_parameters():
    g = sdg[k].grad
    if g is None:
        rows.append((k, None, None, None if p.grad is None else float(p.grad.norm())))
        continue
    h = p.grad.cpu() if p.grad is not None else torch.zeros_like(g)
    ee = float((emu[k].grad - g).norm() / (g.norm() + 1e-30)) if emu is not None and emu[k].grad is not None else None
    rows.append((k, float((h - g).norm() / (g.norm() + 1e-30)), float(h.norm() / (g.norm() + 1e-30)), float(g.norm()), ee))
for row in reversed(rows):
    k, e, ratio, n = row[:4]
    print("%-55s err %-10s ratio %-8s |gref| %-10s emulated %s" % (k, "-" if e is None else "%.4f" % e, "-" if ratio is None else "%.3f" % ratio,
                                                              "-" if n is None else "%.3e" % n, "-" if len(row) < 5 or row[4] is None else "%.4f" % row[4]))
if os.environ.get("CHECKS"):
    from hiputil import op_reference
    from models.dehaze1113 import _plan_backward
    B = _plan_backward(net._plan_for(x.to(DEV)))
    B.checks, B.check_reference = [], op_reference
    net.zero_grad()
    (net(x.to(DEV)) * cot.to(DEV)).sum().backward()
    torch.cuda.synchronize()
    for o in B.checks:
        print("%-34s dw %.4f dx %s" % (o["label"], o["dw"], "%.4f" % o["dx"] if "dx" in o else "-"))
if os.environ.get("SVD"):
    for k in os.environ["SVD"].split(","):
        g, h = sdg[k].grad, dict(net.named_parameters())[k].grad.cpu()
        e = (h - g).reshape(g.shape[0], -1)
        s = torch.linalg.svdvals(e.double())
        print(k, "error singular values (top 4) %s of total %.3e; |gref| %.3e" % ([round(float(v), 4) for v in s[:4]], float(e.norm()), float(g.norm())))
        if g.dim() == 4:
            print("   per-output-channel error norm / ref norm:", [round(float(e[i].norm() / (g[i].norm() + 1e-30)), 3) for i in range(min(8, g.shape[0]))])
            # does the error of filter co look like (constant) x (sum over pixels of the input)?  correlate with the emulated oracle's dW mean structure
            print("   error mean over taps per (co, ci) [first 3 co]:", e.reshape(g.shape[0], g.shape[1], -1).mean(-1)[:3].tolist())

if os.environ.get("DYTAP"):
    from models.dehaze1113 import _plan_backward
    P = net._plan_for(x.to(DEV))
    B = _plan_backward(P)
    rec = [r for r in P.records if r["kind"] == "conv" and r["w"].param is net.conv_refin.weight][0]
    gy = B.G(rec["y"]).torch_nchw().cpu()                     # gradient w.r.t. conv_refin's stored output, as the walk left it
    go = TAPS["x9pre"].grad
    yh = rec["y"].torch_nchw().cpu()
    print("x9pre forward rel-rms %.5f" % float((yh - TAPS["x9pre"].detach()).norm() / TAPS["x9pre"].detach().norm()))
    print("dy rel-rms %.4f   per-channel: sum(ref) / sum|ref| and sum(hip) / sum|hip| (a BatchNorm backward output sums to 0):" % float((gy - go).norm() / go.norm()))
    for c in range(0, gy.shape[1], 3):
        print("   c%-2d ref %+.2e hip %+.2e   mean|dy| %.3e  mean(hip-ref) %+.3e  rms(hip-ref) %.3e" % (
            c, float(go[:, c].sum() / go[:, c].abs().sum()), float(gy[:, c].sum() / gy[:, c].abs().sum()), float(go[:, c].abs().mean()),
            float((gy[:, c] - go[:, c]).mean()), float((gy[:, c] - go[:, c]).pow(2).mean().sqrt())))
    r3 = [r for r in P.records if r["kind"] == "conv" and r["w"].param is net.refine3.weight][0]
    m = r3["pro"]._meta
    xs = r3["x"].torch_nchw().double().cpu()
    mu, sd_ = xs.mean((0, 2, 3)), xs.var((0, 2, 3), unbiased=False).sqrt()
    print("batchnorm20 statistics vs the stored tensor: (mean_stats - mean_stored) / std:", [round(float(v), 6) for v in ((m["mean"].cpu().double()[:xs.shape[1]] - mu) / sd_)[:8]])
    print("   var_stats / var_stored - 1:", [round(float(v), 6) for v in (m["var"].cpu().double()[:xs.shape[1]] / sd_ ** 2 - 1)[:8]])
    xo = TAPS["x9pre"].detach().double()
    print("   (mean_stored - mean_oracle) / std:", [round(float(v), 6) for v in ((mu - xo.mean((0, 2, 3))) / sd_)[:8]])
