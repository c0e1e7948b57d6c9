#!/usr/bin/env python
"""bench.py -- FD-GAN hot path on MI355X.

Workload (BASELINE.json configs[1]): netG (`models.dehaze1113.FDGAN`) forward, bf16
storage / fp32 accumulate, batch 16 @ 3x256x256 per GPU, train-mode BatchNorm (what the
reference runs at inference, README.md:38), synthetic U[0,1) input resident in HBM
before the timed region.  One "step" = one forward of one batch.  N > 1: one process
per GPU, each rank runs its own batch (data parallel, no exchange in the forward path);
value = N*B*K / max-over-ranks time.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline     -- the dominant kernel class (largest share of GPU time): algorithmic
                  bytes (or flops) of its launches / their hipEvent-measured duration
                  inside the timed region, against 8 TB/s HBM or 2.5 PFLOP/s bf16 MFMA.
  cpu_baseline -- the CPU oracle (oracle/, PyTorch-CPU fp32 restatement of the reference,
                  parity-checked against it) timed on this host's cores, N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "fd-gan_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec
MFMA_PEAK_TFLOPS = 2500.0    # dense bf16 MFMA
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
    ap.add_argument("--train", action="store_true",
                    help="NOT the default workload: time the full training step of fd-gan_amd/train.py (G + Fusion-D + VGG16 "
                         "+ SSIM, Adam; BASELINE configs[2], a reconstructed loss composition); no roofline / cpu_baseline")
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


def main():
    a = parse()
    import torch
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    from fdgan_hip.dp import DpContext
    dp = DpContext.from_env(backend="nccl")           # RCCL; one process per GPU
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
    if a.train:
        import train as train_mod
        ts = train_mod.TrainStep(dev, dp=dp)
        gt = torch.from_numpy(np.random.default_rng(99 + rank).random((B, 3, S, S), dtype=np.float32)).to(dev)
        haze = (gt * 0.6 + 0.3).clamp(0, 1)
        for _ in range(max(a.warmup, 1)):
            ts.step(haze, gt)
        barrier()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            last = ts.step(haze, gt)
        barrier()
        dt = dp.max_over_ranks(time.perf_counter() - t0)
        images = dp.sum_over_ranks(B * a.steps)
        if rank == 0:
            print(json.dumps({
                "metric": "training images/sec @256x256 (1/2/4/8 GPUs) + PSNR/SSIM parity on SOTS", "value": round(images / dt, 2),
                "unit": "images/sec", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
                "ms_per_step": round(1e3 * dt / a.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "bf16", "data": "synthetic",
                "config": {"workload": "full training step (fd-gan_amd/train.py): G fwd+bwd, Fusion-D 3 fwd + 3 bwd, VGG16 2 fwd + 1 "
                                       "bwd, SSIM, Adam(G), Adam(D), gradient all-reduce when n_gpus > 1; batch %d @ %dx%d per GPU; "
                                       "loss composition reconstructed (the reference ships no training loop)" % (B, S, S),
                           "global_batch": world * B, "image": [3, S, S], "parallelism": "dp%d" % world,
                           "last_losses": {k: round(v, 4) for k, v in last.items()}},
                "roofline": None, "cpu_baseline": None}), flush=True)
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
                "dtype": "bf16", "data": "synthetic",
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
        headline = classes.get("conv3x3_rs_bn32[128->32 @%dx%d]" % (S, S))
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
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "netG (FDGAN) forward-only, bf16 storage / fp32 accumulate, train-mode "
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
            roof["headline_3x3"] = {"kernel": "conv3x3_rs_bn32[128->32 @%dx%d]" % (S, S), "avg_launch_us": round(hl_us, 2),
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
            with open(a.breakdown, "w") as f:
                json.dump({"total_ms_instrumented": total_ms, "rows": rows}, f, indent=1)
        if world == 1 and not a.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(cpu_sd, S, a.cpu_seconds)
        print(json.dumps(res), flush=True)
    dp.close()


if __name__ == "__main__":
    main()
