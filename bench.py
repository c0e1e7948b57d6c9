#!/usr/bin/env python
"""bench.py -- FD-GAN hot path on MI355X.

Default workload (BASELINE.json configs[2]; configs[3] when N > 1 -- the configurations the metric "training
images/sec @256x256 (1/2/4/8 GPUs)" is quoted on): ONE FULL TRAINING STEP of fd-gan_amd/train.py per "step" -- netG
forward + backward, Fusion-D 3 forwards + 3 backwards, VGG16 2 forwards + 1 backward, SSIM, L1 / MSE / BCE, Adam(D),
Adam(G) -- fp16 forward activations and filters, bf16 gradients, fp32 accumulate (DESIGN.md 'Precision'), batch 16 @ 3x256x256 per GPU, train-mode BatchNorm, synthetic images
resident in HBM before the timed region.  N > 1: one process per GPU, each rank its own batch (weak scaling, global
batch 16 N), RCCL all-reduce of the two flat gradient buffers inside the step; value = N*B*K / max-over-ranks time.

`--forward`: BASELINE.json configs[1] instead (netG forward only; the number DESIGN.md tracks kernel by kernel).  The
default line carries it as the `forward_only` object as well (rank 0, N = 1).

Prints ONE JSON line (rank 0).  Extra objects:
  roofline     -- the dominant kernel of the timed workload (largest share of GPU time in an instrumented warm-up
                  step): algorithmic bytes (or flops) of sampled launches / their hipEvent-measured duration inside
                  the timed region (events on the launch stream, fdgan_kernel_timer_*), against 8 TB/s HBM or
                  2.5 PFLOP/s dense 16-bit MFMA (f16 and bf16 run at the same rate).
  cpu_baseline -- the CPU oracle (oracle/, PyTorch-CPU fp32 restatement of the reference, parity-checked against
                  it) running the same step on this host's cores, bounded sample, N = 1 only.
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on this host driver; before the HIP runtime starts

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "fd-gan_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec
MFMA_PEAK_TFLOPS = 2500.0    # dense bf16 / f16 MFMA (v_mfma_f32_16x16x32_{bf16,f16}: same rate)
RIDGE = MFMA_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=16, help="per-GPU batch (BASELINE config: 16)")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--breakdown", default="", help="write the per-kernel-class breakdown JSON here")
    ap.add_argument("--graph", action="store_true", help="replay the plan as a hipGraph (no per-kernel events)")
    ap.add_argument("--train", action="store_true", help="the default workload (kept for older command lines)")
    ap.add_argument("--forward", action="store_true",
                    help="BASELINE configs[1] instead of the training step: netG forward only, with its per-kernel roofline")
    ap.add_argument("--no-forward-leg", action="store_true", help="skip the forward_only object of the default line")
    ap.add_argument("--no-forward-1024", action="store_true",
                    help="skip the forward_1024 object (tools/pmc_bench.sh: its 4x larger launches of the same kernels would "
                         "pollute per-kernel PMC averages keyed on the B=16 @256^2 shapes)")
    ap.add_argument("--train-g", action="store_true",
                    help="NOT the default workload: time netG forward + backward (mse loss), the part of the training "
                         "step (BASELINE configs[2]) that exists; no roofline / cpu_baseline objects")
    return ap.parse_args()


def classes_of(plan):
    """Group conv launches by (kernel, shape): returns {key: dict(idx=[...], bytes, flops, flops_done)}."""
    names = plan.main.kernel_names()
    out = {}
    for m in plan.meta:
        if not m["launches"] or m["label"] == "op":
            continue
        k0 = m["launches"][0]
        if "cin" in m:
            key = "%s[%d->%d @%dx%d]" % (names[k0], m["cin"], m["cout"], m["h_out"], m["w_out"])
        else:
            key = names[k0]
        c = out.setdefault(key, dict(idx=[], bytes=0.0, flops=0.0, flops_done=0.0, kernel=names[k0]))
        c["idx"].append(k0)
        c["bytes"] += m["bytes"]
        c["flops"] += m["flops"]
        c["flops_done"] += m["flops_done"]
        for extra in m["launches"][1:]:
            e = out.setdefault(names[extra], dict(idx=[], bytes=0.0, flops=0.0, flops_done=0.0, kernel=names[extra]))
            e["idx"].append(extra)
    return out


def cpu_baseline(state_dict, size, seconds):
    """The oracle on the host cores.  Bounded sample: B=1 forwards for ~`seconds`."""
    import torch
    from oracle import dehaze1113_ref as ref
    from oracle.detweights import det_input
    # 16 threads is the measured optimum for a batch-1 forward on the GPU box's 2x64-core host
    # (tools/cpu_threads.py: 8 thr 0.64 s, 16 thr 0.56 s, 32 thr 0.69 s, 128 thr 2.9 s per image)
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    og = ref.FDGAN()
    og.load_state_dict(state_dict)
    x = det_input((1, 3, size, size), seed=1234)
    with torch.no_grad():
        og(x.clone())                                  # warm-up
        n, t0 = 0, time.perf_counter()
        while True:
            og(x.clone())
            n += 1
            dt = time.perf_counter() - t0
            if dt >= seconds or n >= 50:
                break
    return {"value": round(n / dt, 3), "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d netG forwards, batch 1 @%dx%d, fp32 PyTorch-CPU oracle (oracle/dehaze1113_ref.py), %.1f s"
                      % (n, size, size, dt)}


def cpu_baseline_train(size, seconds):
    """The oracle training step (oracle/train_ref.py) on the host cores.  Bounded sample: batch-1 steps for ~`seconds`."""
    import torch
    from oracle.train_ref import TrainStepRef
    from oracle.detweights import det_input
    cores = min(os.cpu_count() or 1, 16)               # see cpu_baseline: the measured optimum for batch 1 on the GPU box's host
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    ts = TrainStepRef()
    gt = det_input((1, 3, size, size), seed=99)
    haze = (gt * 0.6 + 0.3).clamp(0, 1)
    ts.step(haze, gt)                                  # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        ts.step(haze, gt)
        n += 1
        dt = time.perf_counter() - t0
        if dt >= seconds or n >= 20:
            break
    res = {"value": round(n / dt, 3), "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
           "sample": "%d full training steps (G + Fusion-D + VGG16 + SSIM, Adam), batch 1 @%dx%d, fp32 PyTorch-CPU oracle "
                     "(oracle/train_ref.py), %.1f s" % (n, size, size, dt), "host_cpus": os.cpu_count()}
    # SURVEY 8(d) asks for batch 1 AND the benchmark's batch 16: one step of the same oracle (many more threads are SLOWER for
    # torch's CPU convolutions on this host)
    # (round 6, tools/cpu_threads.py train16 on the GPU box's 2 x 64-core host: 16 threads 44.3 s, 32 threads 41.5 s, 64 threads 56.2 s
    # per batch-16 step: the batch-16 leg has its own optimum -- profiles/r6_cpu_threads_train16.txt)
    cores16 = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores16)
    gt16 = det_input((16, 3, size, size), seed=98)
    haze16 = (gt16 * 0.6 + 0.3).clamp(0, 1)
    t0 = time.perf_counter()
    ts.step(haze16, gt16)
    dt16 = time.perf_counter() - t0
    res["batch16"] = {"value": round(16 / dt16, 3), "unit": "images/sec", "cores": cores16,
                      "sample": "1 full training step, batch 16 @%dx%d, same oracle, %.1f s" % (size, size, dt16)}
    return res


# Algorithmic HBM bytes / flops of one launch of the kernels that can dominate the training step, from the arguments of
# the engine call that issues it (exactly one launch of that name per call): (launcher name, bytes, flops).
def _model_bn_bwd_apply(args, kw):
    dpre_fd, accumulate = args[0], (args[6] if len(args) > 6 else kw.get("accumulate", False))
    px = dpre_fd.n * dpre_fd.h * dpre_fd.w
    return "bn_bwd_apply", px * dpre_fd.c * 2 * (4 if accumulate else 3), 0.0      # read dpre, x (and dx), write dx; bf16


def _model_affine_accumulate(args, kw):
    x_fd = args[0]
    return "affine_accumulate", x_fd.n * x_fd.h * x_fd.w * x_fd.c * 2 * 3, 0.0      # read x, dx; write dx


def _model_conv_bwd_data(args, kw):
    """fdgan_conv2d_bwd_data: read dy and the forward input x (mask), then either write dpre or read + write the gradient
    buffer (accumulate); 2 * P * Cin * Cout * k * k flops.  Launcher name as conv_k{1,3,4}.hip pick it."""
    dy_fd, fwd_x_fd, dpre_fd, desc = args[0], args[2], args[4], args[5]
    accumulate = args[7] if len(args) > 7 else kw.get("accumulate", False)
    px = dpre_fd.n * dpre_fd.h * dpre_fd.w
    byts = dy_fd.n * dy_fd.h * dy_fd.w * dy_fd.c * 2 + px * dpre_fd.c * 2 * (3 if accumulate else 2)
    flops = 2.0 * px * dpre_fd.c * dy_fd.c * desc.ksize * desc.ksize
    name = "conv%dx%d_bn%d_bwd" % (desc.ksize, desc.ksize, 32 if dpre_fd.c <= 32 else 128)
    if desc.ksize == 1 and dy_fd.c in (32, 64, 128) and px % 64 == 0 and dpre_fd.c <= 1024:
        name = "conv1x1_bwd_stream"                       # conv1x1_bwd_fits (whole-buffer views in the training step)
    if desc.ksize == 3 and desc.pad == 1 and dy_fd.c == 32 and dpre_fd.c == 128:
        name = "conv3x3_bwd_stream"                       # conv3x3_bwd_fits
    return name, byts, flops


def _model_conv1x1_bwd_data_weight(args, kw):
    """fdgan_conv1x1_bwd_data_weight (the dense-layer bottleneck, data + weight gradient in one pass): the data gradient's
    bytes -- read dy and x, read + write the gradient buffer -- plus, when the pending BatchNorm remainder of dy is applied on
    the fly (dy_affine), the 128-channel activation it multiplies; nothing for the weight gradient, whose operands are on chip.
    (Its [128][C] fp32 partial per workgroup, 64 KB against ~3 MB of activations, is left out like every filter.)  Flops of both."""
    dy_fd, fwd_x_fd, dpre_fd, accumulate = args[0], args[2], args[4], args[6]
    px = dpre_fd.n * dpre_fd.h * dpre_fd.w
    byts = px * dy_fd.c * 2 * (2 if kw.get("dy_affine") is not None else 1) + px * dpre_fd.c * 2 * (3 if accumulate == 1 else 2)
    # the same in whole 128-byte lines (64 channels): x and G are channel PREFIXES of a wider pixel-major buffer, and a row that ends in
    # half a line costs the memory system the whole line (profiles/r6_ubench_raggedrow.txt) -- reported beside the algorithmic figure
    c_lines = (dpre_fd.c + 63) // 64 * 64 if dpre_fd.stride[2] > dpre_fd.c else dpre_fd.c
    lines = px * dy_fd.c * 2 * (2 if kw.get("dy_affine") is not None else 1) + px * c_lines * 2 * (3 if accumulate == 1 else 2)
    return "conv1x1_bwd_wgrad_stream", byts, 2 * 2.0 * px * dpre_fd.c * dy_fd.c, lines


MODELS = {"bn_bwd_apply": ("bn_bwd_apply", _model_bn_bwd_apply), "affine_accumulate": ("affine_accumulate", _model_affine_accumulate),
          "conv1x1_bwd_wgrad_stream": ("conv1x1_bwd_data_weight", _model_conv1x1_bwd_data_weight)}
for _k in (1, 3, 4):
    for _w in (32, 128):
        MODELS["conv%dx%d_bn%d_bwd" % (_k, _k, _w)] = ("conv_bwd_data", _model_conv_bwd_data)
MODELS["conv1x1_bwd_stream"] = ("conv_bwd_data", _model_conv_bwd_data)
MODELS["conv3x3_bwd_stream"] = ("conv_bwd_data", _model_conv_bwd_data)


def train_bench(a, dp, dev, B, S):
    import numpy as np
    import torch
    import train as train_mod
    from fdgan_hip import engine as E
    world, rank = dp.world, dp.rank
    ts = train_mod.TrainStep(dev, dp=dp, synthetic=True)     # random-init weights of the reference architecture (no network for checkpoints)
    gt = torch.from_numpy(np.random.default_rng(99 + rank).random((B, 3, S, S), dtype=np.float32)).to(dev)
    haze = (gt * 0.6 + 0.3).clamp(0, 1)
    for _ in range(max(a.warmup, 1)):
        ts.step(haze, gt)
    torch.cuda.synchronize()
    # ---- one instrumented step (every launch bracketed) -> GPU time per launcher name
    E.kernel_timer_arm(None, 1, 4096)
    ts.step(haze, gt)
    torch.cuda.synchronize()
    samples, n_launch = E.kernel_timer_read(4096)
    by_name = {}
    for _, ms, name in samples:
        d = by_name.setdefault(name, [0, 0.0])
        d[0] += 1
        d[1] += ms
    lib_ms = sum(v[1] for v in by_name.values())
    ranking = sorted(by_name.items(), key=lambda kv: -kv[1][1])
    if a.breakdown and rank == 0:       # every launcher of the instrumented step: launches, GPU ms (hipEvent pairs on the launch stream)
        os.makedirs(os.path.dirname(os.path.abspath(a.breakdown)), exist_ok=True)
        with open(a.breakdown, "w") as f:
            json.dump({"library_gpu_ms_per_step": lib_ms, "launches_per_step": n_launch,
                       "rows": [{"launcher": n, "launches": v[0], "ms": round(v[1], 4), "share": round(v[1] / lib_ms, 4)}
                                for n, v in ranking]}, f, indent=1)
    dom_name = next((n for n, _ in ranking if n in MODELS), None)   # the top kernel with a byte model (see `roofline.ranking`)
    # ---- per-call algorithmic bytes of the dominant kernel, recorded by wrapping its engine entry point
    per_call = []
    if dom_name is not None:
        fn_name, model = MODELS[dom_name]
        orig = getattr(E, fn_name)

        def wrapped(*args, **kw):
            mres = model(args, kw)
            nm, byts_, flops_ = mres[:3]
            res = orig(*args, **kw)
            launched = not (fn_name == "conv1x1_bwd_data_weight" and res is None)   # None: outside the fused kernel, nothing launched
            if nm == dom_name and launched:
                per_call.append((byts_, flops_, mres[3] if len(mres) > 3 else byts_))
            return res
        # The reverse walks are recorded and replayed (fdgan_hip/backward.py: _Tape), so the Python entry point is not called in
        # the timed steps: ONE eager step (recording switched off) lists the kernel's launches of a step in launch order; every
        # step launches the same sequence, so launch i of the timed region has the shape of entry i mod launches-per-step.
        from fdgan_hip import backward as BW
        setattr(E, fn_name, wrapped)               # backward.py looks the function up on the module at call time
        BW.FORCE_EAGER = True
        try:
            ts.step(haze, gt)
            torch.cuda.synchronize()
        finally:
            BW.FORCE_EAGER = False
            setattr(E, fn_name, orig)
        per_step = list(per_call)
    if dom_name is not None:
        import math
        cnt = by_name[dom_name][0]
        stride = max(1, cnt // 8)                                   # ~8 bracketed launches per step ...
        while math.gcd(stride, cnt) != 1:                           # ... walking through every position of the step over the run
            stride += 1
        E.kernel_timer_arm(dom_name, stride, min(65536, 16 * a.steps + 16))
    comm = None
    if world > 1 or dp.force_exchange:
        # the gradient exchange on its own (nothing to hide behind): both flat buffers, same slicing as the step uses
        import torch.distributed as dist
        dp.barrier()
        iso = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ts.optD.allreduce_grads(dp)
            ts.optG.allreduce_grads(dp)
            e1.record()
            torch.cuda.synchronize()
            iso.append(e0.elapsed_time(e1))
        ts.optG.comm_events, ts.optD.comm_events = [], []
        comm = {"rccl_ranks": world, "backend": dist.get_backend(), "bytes_per_step": 4 * (ts.optG.grad.numel() + ts.optD.grad.numel()),
                "isolated_ms_per_step": round(min(iso), 3)}
    dp.barrier()                                   # torch.cuda.synchronize() + a collective barrier when world > 1
    t0 = time.perf_counter()
    for _ in range(a.steps):
        last_dev = ts.step(haze, gt, sync=False)   # the losses stay on the device: no host round trip inside the timed region
    t_enq = time.perf_counter() - t0               # host time to ENQUEUE the steps: close to the total = the host is the limit
    torch.cuda.synchronize()
    dt_rank = time.perf_counter() - t0             # this rank alone (before the closing barrier)
    dp.barrier()
    dt = dp.max_over_ranks(time.perf_counter() - t0)
    images = dp.sum_over_ranks(B * a.steps)
    last = ts.losses_dict(last_dev)
    per_rank_ms = dp.gather_floats(1e3 * dt_rank / a.steps)
    if comm is not None:
        exposed = sum(e0.elapsed_time(e1) for e0, e1 in ts.optG.comm_events + ts.optD.comm_events) / a.steps
        comm["exposed_ms_per_step"] = round(exposed, 3)       # stream stalled on the collectives (G: after its backward; D: on the side stream, beside VGG16)
        comm["ms_per_step_by_rank"] = [round(v, 3) for v in per_rank_ms]
        comm["hidden_fraction"] = round(max(0.0, 1.0 - exposed / max(comm["isolated_ms_per_step"], 1e-9)), 3)
        ts.optG.comm_events = ts.optD.comm_events = None
    # ---- how long the HOST needs for a step when nothing holds it back.  Inside the timed loop the runtime's queue fills up and
    # hipLaunchKernel blocks, so there the enqueue time is just the GPU time again (`..._backpressured`); a step enqueued onto
    # an IDLE GPU (synchronize first) is what the host itself costs -- recorded plans replayed, loss plumbing, Adam launches.
    host_idle = []
    for _ in range(5):
        torch.cuda.synchronize()
        th = time.perf_counter()
        ts.step(haze, gt, sync=False)
        host_idle.append(time.perf_counter() - th)
    torch.cuda.synchronize()
    host_idle.sort()
    roof = None
    if dom_name is not None:
        timed, seen = E.kernel_timer_read(65536)
        seen -= len(host_idle) * by_name[dom_name][0]          # the five extra steps above launched the kernel as well (never sampled: the pool was full or the index is past the timed steps)
        timed = [t for t in timed if t[0] < by_name[dom_name][0] * a.steps]
        n_ps = len(per_step)
        if n_ps == 0 or n_ps != by_name[dom_name][0] or seen != n_ps * a.steps:      # the model's launcher-name guess disagreed
            timed, per_step, n_ps = [], [(0.0, 0.0, 0.0)], 1                            # with the dispatch: no roofline rather than a wrong one
        per_call = per_step
        byts = sum(per_step[i % n_ps][0] for i, _, _ in timed)
        flops = sum(per_step[i % n_ps][1] for i, _, _ in timed)
        lines_b = sum(per_step[i % n_ps][2] for i, _, _ in timed)
        t_ms = max(sum(ms for _, ms, _ in timed), 1e-9)
        n_step = by_name[dom_name][0]
        if flops / max(byts, 1.0) < RIDGE:
            ach = byts / (t_ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4)}
        else:
            ach = flops / (t_ms * 1e-3) / 1e12
            roof = {"bound": "mfma", "achieved": round(ach, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / MFMA_PEAK_TFLOPS, 4)}
        roof.update({"traffic": None, "kernel": dom_name, "launches_per_step": n_step, "launches_timed": len(timed),
                     "avg_launch_us": round(t_ms / max(len(timed), 1) * 1e3, 2),
                     "algorithmic_mb_per_launch": round(sum(c[0] for c in per_call) / max(len(per_call), 1) / 1e6, 2),
                     "gflop_per_launch": round(sum(c[1] for c in per_call) / max(len(per_call), 1) / 1e9, 2),
                     "tflops": round(flops / (t_ms * 1e-3) / 1e12, 1),
                     "share_of_library_gpu_time": round(by_name[dom_name][1] / lib_ms, 3),
                     "ranking_ms_per_step": {n: round(v[1], 3) for n, v in ranking[:8]}})
        if lines_b > byts:   # informative, never `frac`: the same launches priced in the 128-byte lines their channel prefixes touch
            roof["line_granular"] = {"mb_per_launch": round(sum(c[2] for c in per_call) / max(len(per_call), 1) / 1e6, 2),
                                     "GB/s": round(lines_b / (t_ms * 1e-3) / 1e9, 1), "frac": round(lines_b / (t_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                     "note": "x / G rows are channel prefixes of 64k + 32 channels in half of the dense layers: a row ending in half a "
                                             "128-byte line costs the whole line (profiles/r6_ubench_raggedrow.txt); `achieved` / `frac` stay algorithmic"}
        try:   # HBM bytes of this kernel from the committed PMC passes (same workload), else null
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                pmc = json.load(f).get("%s@train_B%d_%d" % (dom_name, B, S))
            if pmc:
                traffic_mb = (2.0 * pmc["fetch_kib"] + pmc["write_kib"]) * 1024 / 1e6
                roof["traffic_mb_per_launch"] = round(traffic_mb, 2)
                if roof["unit"] == "GB/s" and roof["algorithmic_mb_per_launch"] > 0:
                    traffic = ach * traffic_mb / roof["algorithmic_mb_per_launch"]
                    # a PMC row that mixes launches of other shapes would price traffic above the pins: not evidence
                    roof["traffic"] = round(traffic, 1) if traffic <= HBM_PEAK_GBS else None
        except (OSError, ValueError):
            pass
    # ---- whole-step roofline: counted work of the step / measured step time.  Work is counted from the plans' own op logs
    # (NetPlan.meta: 2*MACs of every conv on the reference's formulation; bytes = every conv reads its input once and
    # writes its output once in bf16).  Multipliers: a backward is one data-gradient + one weight-gradient
    # pass, each with the forward's MACs and (input + output) bytes; frozen networks get the data gradient only:
    #   G: fwd + dgrad + wgrad = 3;  D: 3 fwd + 2 x (dgrad + wgrad) [D step] + 1 x dgrad [G step] = 8;  VGG16: 2 fwd + dgrad = 3
    step_roof = None
    try:
        with torch.no_grad():
            pg = ts.netG.hip_plan(haze)
            pd = ts.netD.hip_plan(torch.empty(B, 9, S, S, device=dev))
            pv = ts.vgg._plan_for(haze, 0)
        work = {}
        for nm, pl, mult in (("netG", pg, 3), ("netD", pd, 8), ("vgg16", pv, 3)):
            fl = sum(m["flops"] for m in pl.meta)
            by = sum(m["bytes"] for m in pl.meta)
            work[nm] = {"fwd_gflop": round(fl / 1e9, 1), "fwd_algorithmic_gb": round(by / 1e9, 3), "passes": mult}
        gflop = sum(w["fwd_gflop"] * w["passes"] for w in work.values())
        gb = sum(w["fwd_algorithmic_gb"] * w["passes"] for w in work.values())
        ms_step = 1e3 * dt / a.steps
        step_roof = {"gflop_per_step": round(gflop, 1), "algorithmic_gb_per_step": round(gb, 2),
                     "tflops": round(gflop / ms_step, 1), "frac_mfma": round(gflop / ms_step / MFMA_PEAK_TFLOPS, 4),
                     "GB/s": round(gb / ms_step * 1e3, 1), "frac_hbm": round(gb / ms_step * 1e3 / HBM_PEAK_GBS, 4),
                     "arithmetic_intensity": round(gflop / gb, 1), "ridge": round(RIDGE, 1), "networks": work,
                     "tflop_per_image": round(gflop / 1e3 / B, 3)}
    except Exception as e:      # measurement aid only: never fail the benchmark line
        step_roof = {"error": str(e)[:200]}
    res = None
    if rank == 0:
        res = {"metric": "training images/sec @256x256 (1/2/4/8 GPUs) + PSNR/SSIM parity on SOTS", "value": round(images / dt, 2),
               "unit": "images/sec", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
               "ms_per_step": round(1e3 * dt / a.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f16+bf16", "data": "synthetic",
               "config": {"workload": "full training step (fd-gan_amd/train.py; BASELINE.json configs[%d]): G fwd+bwd, Fusion-D 3 fwd + 3 "
                                      "bwd, VGG16 2 fwd + 1 bwd, SSIM, Adam(G), Adam(D), gradient all-reduce when n_gpus > 1; batch %d @ "
                                      "%dx%d per GPU; loss composition reconstructed (the reference ships no training loop)"
                                      % (2 if world == 1 else 3, B, S, S),
                          "global_batch": world * B, "image": [3, S, S], "parallelism": "dp%d" % world,
                          "library_launches_per_step": n_launch, "library_gpu_ms_per_step_instrumented": round(lib_ms, 2),
                          "host_enqueue_ms_per_step": round(1e3 * host_idle[len(host_idle) // 2], 3),
                          "host_enqueue_ms_per_step_backpressured": round(1e3 * t_enq / a.steps, 3),
                          "gradient_exchange": comm,
                          "last_losses": {k: round(v, 4) for k, v in last.items()}},
               "roofline": roof, "step_roofline": step_roof, "cpu_baseline": None}
    del ts
    torch.cuda.empty_cache()
    return res


def forward_1024(g, dev, batch=4, size=1024, warm=2, steps=5):
    """BASELINE.json configs[4]: netG inference at batch 4 @ 1024x1024 (the reference's demo.py defaults imageSize to 1024),
    train-mode BatchNorm as the reference runs it.  Attached to the default line so the driver times it too."""
    import numpy as np
    import torch
    x = torch.from_numpy(np.random.default_rng(4321).random((batch, 3, size, size), dtype=np.float32)).to(dev)
    with torch.no_grad():
        for _ in range(warm):
            y = g(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            y = g(x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        plan = g.hip_plan(x)
    fl, by = sum(m["flops"] for m in plan.meta), sum(m["bytes"] for m in plan.meta)
    ok = bool(torch.isfinite(y).all())
    del y
    torch.cuda.empty_cache()
    return {"value": round(batch / dt, 2), "unit": "images/sec", "ms_per_step": round(1e3 * dt, 3), "steps": steps,
            "workload": "netG (FDGAN) forward-only, batch %d @ %dx%d, fp16 storage / fp32 accumulate, train-mode BatchNorm "
                        "(BASELINE.json configs[4])" % (batch, size, size),
            "gflop_per_step": round(fl / 1e9, 1), "algorithmic_gb_per_step": round(by / 1e9, 2),
            "tflops": round(fl / dt / 1e12, 1), "frac_mfma": round(fl / dt / 1e12 / MFMA_PEAK_TFLOPS, 4),
            "GB/s": round(by / dt / 1e9, 1), "frac_hbm": round(by / dt / 1e9 / HBM_PEAK_GBS, 4), "finite": ok}


def freqsplit_1024(dev, batch=4, size=1024, iters=20):
    """BASELINE.json north_star's frequency split at configs[4]'s size: Blur, Laplacian and the Fusion-discriminator's input
    (cat([img, LF, HF]): train.py's fusion_input) on batch 4 x 3 x 1024 x 1024 fp32 planes, each timed with stream events over
    `iters` launches.  Algorithmic bytes: one read + one write of the tensor for a filter, one read + three writes for the
    concatenation (kernel table)."""
    import torch
    from fdgan_hip import engine as E
    from loss import fusion_input
    x = torch.rand(batch, 3, size, size, device=dev)
    tensor_bytes = x.numel() * 4

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3

    out = {"workload": "Blur(15, sigma 3) / Laplacian(3) / cat([img, LF, HF]) on %d x 3 x %d x %d fp32" % (batch, size, size)}
    with torch.no_grad():
        for name, fn, nbytes in (("blur15", lambda: E.blur15(x, True), 2 * tensor_bytes), ("laplacian3", lambda: E.laplacian3(x), 2 * tensor_bytes),
                                 ("fusion_input", lambda: fusion_input(x), 4 * tensor_bytes)):
            us = timed(fn)
            out[name] = {"us": round(us, 1), "GB/s": round(nbytes / us / 1e3, 1), "frac_hbm": round(nbytes / us / 1e3 / HBM_PEAK_GBS, 4)}
    del x
    torch.cuda.empty_cache()
    return out


def self_launch(a):
    """`python bench.py --gpus N` with no launcher around it (WORLD_SIZE unset): start the N ranks ourselves, one process
    per GPU, exactly as the driver's `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ...` would, and
    pass rank 0's JSON line through.  The reference's multi-GPU mechanism is nn.DataParallel(netG) inside ONE process
    (/root/reference/demo.py:89); one process per GPU over RCCL is this build's form of it."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "8"))
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(a)
    import torch
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    from fdgan_hip.dp import DpContext
    if os.environ.get("FDGAN_BENCH_SHARED_GPU") == "1":
        # test hook (tests/test_dp_step_gpu.py): all ranks on cuda:0 over gloo, to exercise the N > 1 code path of this
        # file on a one-GPU box (RCCL refuses two ranks on one device).  Never set by the driver.
        dp = DpContext.from_env(backend="gloo", device=torch.device("cuda", 0))
    else:
        dp = DpContext.from_env(backend="nccl")       # RCCL; one process per GPU
    world, rank, dev = dp.world, dp.rank, dp.device
    if a.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch N>1 with `python -m torch.distributed.run "
                         "--nproc-per-node N ... bench.py --gpus N`" % (a.gpus, world))

    from fdgan_hip import lib
    lib.load()
    import models.dehaze1113 as net
    import numpy as np

    torch.manual_seed(0)
    g = net.FDGAN()                                   # random-init weights of the reference architecture
    cpu_sd = {k: v.clone() for k, v in g.state_dict().items()}
    g = g.to(dev)
    B, S = a.batch, a.size
    x = torch.from_numpy(np.random.default_rng(1234 + rank).random((B, 3, S, S), dtype=np.float32)).to(dev)

    barrier = dp.barrier
    train_res = None
    if not a.forward and not a.train_g:
        train_res = train_bench(a, dp, dev, B, S)
        if a.no_forward_leg or world != 1:
            if rank == 0:
                if world == 1 and not a.no_cpu_baseline:
                    train_res["cpu_baseline"] = cpu_baseline_train(S, a.cpu_seconds)
                print(json.dumps(train_res), flush=True)
            dp.close()
            return
    if a.train_g:
        tgt = torch.rand(B, 3, S, S, device=dev) * 2 - 1

        def gstep():
            g.zero_grad(set_to_none=True)
            ((g(x) - tgt) ** 2).mean().backward()
        for _ in range(max(a.warmup, 1)):
            gstep()
        barrier()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            gstep()
        barrier()
        dt = dp.max_over_ranks(time.perf_counter() - t0)
        images = dp.sum_over_ranks(B * a.steps)
        if rank == 0:
            print(json.dumps({
                "metric": "training images/sec @256x256 (1/2/4/8 GPUs) + PSNR/SSIM parity on SOTS", "value": round(images / dt, 2),
                "unit": "images/sec", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
                "ms_per_step": round(1e3 * dt / a.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "f16+bf16", "data": "synthetic",
                "config": {"workload": "PARTIAL training step: netG (FDGAN) forward + backward, mse loss, train-mode BatchNorm, "
                                       "batch %d @ %dx%d per GPU; no D / VGG / optimizer / all-reduce yet" % (B, S, S),
                           "global_batch": world * B, "image": [3, S, S], "parallelism": "dp%d" % world},
                "roofline": None, "cpu_baseline": None}), flush=True)
        dp.close()
        return

    with torch.no_grad():
        for _ in range(max(a.warmup, 1)):
            y = g(x)
        barrier()
        plan = g.hip_plan(x)
        classes = classes_of(plan)
        # one instrumented replay (event after every launch) to find the dominant class
        ms = plan.main.profile()
        for c in classes.values():
            c["ms"] = sum(ms[i] for i in c["idx"])
        total_ms = sum(ms)
        # The roofline object is per KERNEL (all of its launches in a step, whatever their shapes), so that
        # its average launch duration is the number a rocprofv3 --stats summary shows for that kernel.
        kernels = {}
        for k, c in classes.items():
            if c["bytes"] > 0 and c["flops"] > 0:
                d = kernels.setdefault(c["kernel"], dict(idx=[], bytes=0.0, flops=0.0, flops_done=0.0, ms=0.0,
                                                         kernel=c["kernel"], shapes=[]))
                d["idx"] += c["idx"]
                d["shapes"].append(k)
                for f in ("bytes", "flops", "flops_done", "ms"):
                    d[f] += c[f]
        dom_key = max(kernels, key=lambda k: kernels[k]["ms"])
        dom = kernels[dom_key]
        # Event pairs cost a few microseconds each: bracket an evenly spread sample (<= 8 launches per step) of
        # the dominant kernel's launches inside the timed region and price exactly those launches.
        per_launch = {m["launches"][0]: m for m in plan.meta if m["launches"] and "cin" in m}
        by_ms = sorted(dom["idx"], key=lambda i: ms[i])           # quantile midpoints: sample mean ~ population mean
        nq = min(8, len(by_ms))
        sample = sorted({by_ms[min(len(by_ms) - 1, int((q + 0.5) * len(by_ms) / nq))] for q in range(nq)})
        dom = dict(dom, idx=sample, bytes_all=dom["bytes"], flops_all=dom["flops"], bytes=sum(per_launch[i]["bytes"] for i in sample),
                   flops=sum(per_launch[i]["flops"] for i in sample), ms=sum(ms[i] for i in sample),
                   ms_all=dom["ms"], launches_all=len(dom["idx"]))
        headline = classes.get("conv3x3_rs2_bn32[128->32 @%dx%d]" % (S, S)) or classes.get("conv3x3_rs_bn32[128->32 @%dx%d]" % (S, S))
        if a.graph:
            plan.main.instantiate_graph()
        else:
            plan.main.time_launches(dom["idx"])
            plan.main.read_timing()
        barrier()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            y = g(x)
        barrier()
        dt = time.perf_counter() - t0
    assert bool(torch.isfinite(y).all())
    dt = dp.max_over_ranks(dt)
    images = dp.sum_over_ranks(B * a.steps)

    if rank == 0:
        res = {
            "metric": "training images/sec @256x256 (1/2/4/8 GPUs) + PSNR/SSIM parity on SOTS",
            "value": round(images / dt, 2), "unit": "images/sec", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(1e3 * dt / a.steps, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f16+bf16", "data": "synthetic",
            "config": {"workload": "netG (FDGAN) forward-only, fp16 storage / fp32 accumulate, train-mode "
                                   "BatchNorm, batch %d @ %dx%d per GPU (BASELINE.json configs[1])" % (B, S, S),
                       "global_batch": world * B, "image": [3, S, S], "parallelism": "dp%d" % world,
                       "launches_per_step": len(plan.main) + 2, "replay": "hipGraph" if a.graph else "eager+events",
                       "gflop_per_step": round(sum(m["flops"] for m in plan.meta) / 1e9, 1),
                       "algorithmic_gb_per_step": round(sum(m["bytes"] for m in plan.meta) / 1e9, 3)},
        }
        if a.graph:
            t_ms, cnt = dom["ms"], len(dom["idx"])            # from the instrumented warm-up replay
        else:
            t_ms, cnt = plan.main.read_timing()
        per_launch_ms = t_ms / max(cnt, 1)
        nl = len(dom["idx"])
        ai = dom["flops"] / dom["bytes"]
        if ai < RIDGE:
            ach = dom["bytes"] / nl / (per_launch_ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 4)}
        else:
            ach = dom["flops"] / nl / (per_launch_ms * 1e-3) / 1e12
            roof = {"bound": "mfma", "achieved": round(ach, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / MFMA_PEAK_TFLOPS, 4)}
        traffic, traffic_mb = None, None
        try:   # HBM bytes of this kernel from the committed PMC passes (same shape and batch), else null
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                pmc = json.load(f).get("%s@netG_B%d_%d" % (dom_key, B, S))
            if pmc:
                traffic_mb = (2.0 * pmc["fetch_kib"] + pmc["write_kib"]) * 1024 / 1e6
                # same unit as `achieved`: PMC bytes per launch (all launches of the kernel) priced at the achieved rate
                traffic = round(roof["achieved"] * traffic_mb * 1e6 / (dom["bytes_all"] / dom["launches_all"]), 1) \
                    if roof["unit"] == "GB/s" else None
                if traffic is not None and traffic > HBM_PEAK_GBS:      # a PMC row polluted by launches of another shape: not evidence
                    traffic = None
        except (OSError, ValueError):
            pass
        roof.update({"traffic": traffic, "traffic_mb_per_launch": round(traffic_mb, 2) if traffic_mb else None,
                     "kernel": dom_key, "launches_per_step": dom["launches_all"], "launches_sampled_per_step": nl,
                     "avg_launch_us": round(per_launch_ms * 1e3, 2), "launches_timed": cnt,
                     "algorithmic_mb_per_launch": round(dom["bytes_all"] / dom["launches_all"] / 1e6, 2),
                     "gflop_per_launch": round(dom["flops_all"] / dom["launches_all"] / 1e9, 2),
                     "avg_launch_us_all_launches_instrumented": round(dom["ms_all"] / dom["launches_all"] * 1e3, 2),
                     "tflops": round(dom["flops"] / nl / (per_launch_ms * 1e-3) / 1e12, 1),
                     "share_of_gpu_time": round(dom["ms_all"] / total_ms, 3)})
        if headline is not None and headline["ms"] > 0:   # the north-star shape (3x3 128->32 @256^2), from the instrumented replay
            hl_us = headline["ms"] / len(headline["idx"]) * 1e3
            roof["headline_3x3"] = {"kernel": "conv3x3_rs2_bn32[128->32 @%dx%d]" % (S, S), "avg_launch_us": round(hl_us, 2),
                                    "achieved_GB/s": round(headline["bytes"] / len(headline["idx"]) / (hl_us * 1e-6) / 1e9, 1),
                                    "frac_hbm": round(headline["bytes"] / len(headline["idx"]) / (hl_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                                    "tflops": round(headline["flops"] / len(headline["idx"]) / (hl_us * 1e-6) / 1e12, 1)}
        res["roofline"] = roof
        if a.breakdown:
            rows = []
            for k, c in sorted(classes.items(), key=lambda kv: -kv[1]["ms"]):
                n_l = len(c["idx"])
                rows.append({"class": k, "launches": n_l, "ms": round(c["ms"], 4),
                             "share": round(c["ms"] / total_ms, 4),
                             "GB/s": round(c["bytes"] / (c["ms"] * 1e-3) / 1e9, 1) if c["bytes"] else None,
                             "TFLOP/s": round(c["flops"] / (c["ms"] * 1e-3) / 1e12, 1) if c["flops"] else None})
            os.makedirs(os.path.dirname(os.path.abspath(a.breakdown)), exist_ok=True)
            # attached to the training line: the step's own table already sits in a.breakdown
            path = a.breakdown if train_res is None else os.path.splitext(a.breakdown)[0] + "_forward.json"
            with open(path, "w") as f:
                json.dump({"total_ms_instrumented": total_ms, "rows": rows}, f, indent=1)
        if train_res is not None:       # default line: the training step, with the forward-only measurement attached
            train_res["forward_only"] = {"value": res["value"], "unit": "images/sec", "ms_per_step": res["ms_per_step"],
                                         "workload": res["config"]["workload"], "roofline": res["roofline"]}
            if not a.no_forward_1024:
                train_res["forward_1024"] = forward_1024(g, dev)
                train_res["freqsplit_1024"] = freqsplit_1024(dev)
            if world == 1 and not a.no_cpu_baseline:
                train_res["cpu_baseline"] = cpu_baseline_train(S, a.cpu_seconds)
            print(json.dumps(train_res), flush=True)
        else:
            if world == 1 and not a.no_cpu_baseline:
                res["cpu_baseline"] = cpu_baseline(cpu_sd, S, a.cpu_seconds)
            print(json.dumps(res), flush=True)
    dp.close()


if __name__ == "__main__":
    main()
