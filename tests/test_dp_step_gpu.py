"""Data-parallel TRAINING STEP with two ranks (BASELINE configs[3] in miniature), on the one GPU of the test box: two
processes share cuda:0 and exchange gradients over gloo (RCCL refuses two ranks on one device; the collective's
semantics are the same, and `torch.distributed` is the only thing that differs from the 8-GPU run).

Each rank builds its own TrainStep from a DIFFERENT torch seed and owns its own half of the global batch.  Checked:
  * the initial broadcast makes the replicas identical (parameters and BatchNorm buffers of rank 0 everywhere);
  * what every rank's optimizers apply is exactly the MEAN of the two ranks' local gradients -- each all-reduced slice
    is snapshotted before the collective, the slices tile the flat buffers -- for D (reduced after its two backward
    passes) and for G (slices reduced while the backward walk is still running);
  * the parameters after the step are bit-identical on both ranks and equal Adam's first update on that mean gradient;
  * BatchNorm statistics stay per replica (nn.DataParallel semantics, /root/reference/demo.py:89): running means differ.
"""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "fd-gan_amd")]
    import torch.distributed as dist
    import train as train_mod
    from fdgan_hip.dp import DpContext
    dp = DpContext.from_env(backend="gloo", device=torch.device("cuda", 0))
    torch.manual_seed(100 + rank)                          # replicas start DIFFERENT: the broadcast has to fix that
    ts = train_mod.TrainStep(dp.device, dp=dp, synthetic=True)
    bn = ts.netG.dense_block1.denselayer1.norm1
    init = dict(g=ts.optG.flat.clone().cpu(), d=ts.optD.flat.clone().cpu(), rm=bn.running_mean.clone().cpu())
    g = torch.Generator().manual_seed(7 + rank)
    gt = torch.rand(2, 3, 64, 64, generator=g).to(dp.device)
    haze = (gt * 0.6 + 0.3).clamp(0, 1)
    snaps = {"g": [], "d": []}
    orig = dist.all_reduce

    def recording_all_reduce(t, op=dist.ReduceOp.SUM, async_op=False, **kw):
        for key, opt in (("g", ts.optG), ("d", ts.optD)):
            base, n = opt.grad.data_ptr(), opt.grad.numel()
            if t.is_cuda and base <= t.data_ptr() < base + 4 * n:
                snaps[key].append(((t.data_ptr() - base) // 4, t.detach().clone().cpu()))
        return orig(t, op=op, async_op=async_op, **kw)
    dist.all_reduce = recording_all_reduce
    ts.optG.comm_events, ts.optD.comm_events = [], []
    losses = ts.step(haze, gt)
    dist.all_reduce = orig
    torch.cuda.synchronize()
    torch.save(dict(init=init, snaps=snaps, losses=losses, g_after=ts.optG.flat.cpu(), d_after=ts.optD.flat.cpu(),
                    g_grad=ts.optG.grad.cpu(), d_grad=ts.optD.grad.cpu(), rm_after=bn.running_mean.cpu(),
                    n_events=len(ts.optG.comm_events) + len(ts.optD.comm_events)), os.path.join(out_dir, "r%d.pt" % rank))
    dp.close()


def test_two_rank_training_step_applies_the_mean_gradient(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(os.path.join(str(tmp_path), "r%d.pt" % i)) for i in range(world)]
    # replicas identical after the initial broadcast although the seeds differed
    for k in ("g", "d", "rm"):
        assert torch.equal(r[0]["init"][k], r[1]["init"][k]), k
    for key, lr in (("g", 2e-4), ("d", 2e-4)):
        n = r[0][key + "_grad"].numel()
        local = []
        for x in r:
            full, cover = torch.zeros(n), torch.zeros(n, dtype=torch.bool)
            for off, t in x["snaps"][key]:
                assert not cover[off:off + t.numel()].any()          # every element reduced exactly once
                full[off:off + t.numel()], cover[off:off + t.numel()] = t, True
            assert bool(cover.all())
            local.append(full)
        assert not torch.equal(local[0], local[1])                   # different half-batches: different local gradients
        mean = (local[0] + local[1]) / 2
        for x in r:
            assert torch.allclose(x[key + "_grad"], mean, rtol=1e-6, atol=1e-12), key
        assert torch.equal(r[0][key + "_after"], r[1][key + "_after"]), key          # replicas stay bit-identical
        # Adam's first step on the mean gradient: p - lr * g / (|g| + eps * sqrt(1 - beta2)) (bias-corrected), eps tiny
        p0, gm = r[0]["init"][key], r[0][key + "_grad"]
        m_hat, v_hat = gm, gm * gm
        want = p0 - lr * m_hat / (v_hat.sqrt() + 1e-8)
        assert torch.allclose(r[0][key + "_after"], want, rtol=0, atol=2e-7), float((r[0][key + "_after"] - want).abs().max())
    assert len(r[0]["snaps"]["g"]) >= 2                                # the generator's gradient left in several slices
    assert not torch.equal(r[0]["rm_after"], r[1]["rm_after"])         # BatchNorm statistics are per replica
    assert r[0]["n_events"] == 2 and all(abs(v) < 1e4 for v in r[0]["losses"].values())


def test_bench_py_runs_with_two_ranks(tmp_path):
    """`python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2` end to end (both ranks on cuda:0 over gloo
    through bench.py's FDGAN_BENCH_SHARED_GPU hook): one JSON line from rank 0, whole-job images/s, the gradient-exchange
    object with its isolated / exposed times."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FDGAN_BENCH_SHARED_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--batch", "2", "--size", "64"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["global_batch"] == 4 and d["config"]["parallelism"] == "dp2"
    assert d["value"] == pytest.approx(2 * 2 * 2 / (d["ms_per_step"] * 2 / 1e3), rel=1e-3)        # all ranks' images / slowest rank's time
    ex = d["config"]["gradient_exchange"]
    assert ex["rccl_ranks"] == 2 and ex["bytes_per_step"] > 4e7 and ex["isolated_ms_per_step"] > 0 and 0.0 <= ex["hidden_fraction"] <= 1.0
    assert d["cpu_baseline"] is None and "forward_only" not in d


def test_bench_py_single_rank_through_rccl(tmp_path):
    """`python -m torch.distributed.run --nproc-per-node 1 bench.py --gpus 1` with the `nccl` backend (= RCCL) and
    FDGAN_DP_FORCE_EXCHANGE=1: process-group bootstrap on 127.0.0.1, the barrier / MAX / SUM timing collectives and the
    bucketed gradient all-reduces (overlapped for G, side-stream for D) all run on RCCL with one rank -- the exact code
    path of the 8-GPU run except for the ring itself."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FDGAN_DP_FORCE_EXCHANGE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
           "--batch", "2", "--size", "64", "--no-forward-leg", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["config"]["parallelism"] == "dp1"
    ex = d["config"]["gradient_exchange"]
    assert ex["backend"] == "nccl" and ex["rccl_ranks"] == 1 and ex["bytes_per_step"] > 4e7 and ex["isolated_ms_per_step"] > 0
    assert len(ex["ms_per_step_by_rank"]) == 1 and all(abs(v) < 1e4 for v in d["config"]["last_losses"].values())


@pytest.mark.gpu
@pytest.mark.parametrize("gpus", [1, 2, 8])
def test_bench_py_plain_python_launches_its_own_ranks(gpus):
    """The driver's command shape with NO launcher around it: `python bench.py --gpus N --steps K --warmup W` and WORLD_SIZE
    unset.  N = 1 runs in-process; N = 2 must start its own two ranks (bench.self_launch -> torch.distributed.run on
    127.0.0.1; both on cuda:0 over gloo through the FDGAN_BENCH_SHARED_GPU hook on this one-GPU box) instead of dying on an
    argument check (VERDICT r3, next #1a).  The reference's only multi-GPU mechanism is /root/reference/demo.py:89.
    N = 8 (VERDICT r4, next #7c) is configs[3]'s rank count as a dry run at toy size: eight ranks' rendezvous, replica sync,
    slice-wise all-reduce and per-rank step times, so that the first real 8-GPU lease does not meet an untested world size."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["FDGAN_BENCH_SHARED_GPU"] = "1"
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", str(gpus), "--steps", "2", "--warmup", "1",
           "--batch", "2", "--size", "64", "--no-forward-leg", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    if out.returncode != 0 and gpus > 1 and "HSA_STATUS_ERROR_ILLEGAL_INSTRUCTION" in out.stderr and "sync_replicas" in out.stderr:
        # Round 6, profiles/r6_oversubscription.txt: with EIGHT processes on ONE GPU (this test's hook only -- RCCL refuses it and the
        # driver never does it) about one launch in 30 loses a rank to an illegal-instruction queue abort while the replicas are
        # being synchronised: every launch of this library in the dying rank had completed (the gloo broadcasts in front of that
        # point synchronise the stream), only PyTorch's checksum kernels / gloo's copies were in flight, and 14 000 steady-state
        # iterations + 320 process start-ups of this library's kernels alone under the same oversubscription lost none.  Start-up
        # markers pinned it to PyTorch's fp64 checksum kernels on a drained device; those checksums are formed on the host since
        # (64 of 64 launches clean).  One retry stays, recorded; a second abort fails the test.
        try:
            with open(os.path.join(root, "gpurun_out", "bench_ranks%d_startup_abort.txt" % gpus), "w") as f:
                f.write(out.stderr)
        except OSError:
            pass
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    if out.returncode != 0:      # the launcher's summary (SIGTERMs of the surviving ranks) hides the first failure: show the tracebacks
        err = out.stderr.splitlines()
        first = [i for i, ln in enumerate(err) if "Traceback" in ln or "Error" in ln][:6]
        excerpt = "\n".join("\n".join(err[max(0, i - 2):i + 12]) for i in first[:3])
        try:
            with open(os.path.join(root, "gpurun_out", "bench_ranks%d_stderr.txt" % gpus), "w") as f:
                f.write(out.stderr)
        except OSError:
            pass
        assert False, excerpt[-6000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == gpus and d["config"]["parallelism"] == "dp%d" % gpus and d["config"]["global_batch"] == 2 * gpus
    assert d["steps"] == 2 and d["warmup"] == 1 and d["value"] > 0
    if gpus > 1:
        assert d["config"]["gradient_exchange"]["rccl_ranks"] == gpus


@pytest.mark.gpu
def test_c_abi_allreduce_one_rank_communicator():
    """include/fdgan_hip.h fdgan_allreduce_*: the RCCL wrapper a non-PyTorch host would use for the gradient exchange,
    as far as ONE GPU lets it be exercised: unique id, a world-size-1 communicator, an in-place fp32 sum on a side stream (the
    identity at world 1), destroy; and the error path (bad rank) through fdgan_last_error."""
    import ctypes as C
    import torch
    from fdgan_hip import lib as L
    lib = L.load()
    ident = C.create_string_buffer(128)
    L.check(lib.fdgan_allreduce_unique_id(ident), "allreduce_unique_id")
    assert any(ident.raw), "empty unique id"
    comm = C.c_void_p()
    assert lib.fdgan_allreduce_comm_create(ident, 3, 2, C.byref(comm)) < 0 and b"rank" in lib.fdgan_last_error()
    L.check(lib.fdgan_allreduce_comm_create(ident, 0, 1, C.byref(comm)), "allreduce_comm_create")
    assert comm.value
    dev = torch.device("cuda", 0)
    buf = torch.arange(1 << 20, dtype=torch.float32, device=dev) * 0.5
    want = buf.clone()
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        L.check(lib.fdgan_allreduce_sum_f32(comm, buf.data_ptr(), buf.numel(), side.cuda_stream), "allreduce_sum_f32")
    side.synchronize()
    assert torch.equal(buf, want)
    L.check(lib.fdgan_allreduce_comm_destroy(comm), "allreduce_comm_destroy")


def test_train_py_real_data_loop_single_and_two_ranks(tmp_path):
    """`train.py --dataroot` end to end on .h5 pairs (the reference's data format, /root/reference/datasets/pix2pix.py:62-77): one
    process (the reference's loader as it is: shuffled, ragged last batch), then TWO ranks sharing this GPU over gloo on a dataset
    whose batch count is odd per rank pair -- the ADVICE r3 deadlock scenario on the real loop: both ranks must finish both epochs
    (fdgan_hip.dp.RankBatches: equal step counts), rank 0 writes `module.`-prefixed checkpoints that demo.py's loader accepts."""
    import subprocess
    import sys
    import numpy as np
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [os.path.join(root, "fd-gan_amd")]
    from datasets.pix2pix import write_pair
    data = str(tmp_path / "data")
    rng = np.random.default_rng(3)
    for i in range(7):
        gt = rng.random((64, 64, 3), dtype=np.float32)
        write_pair(data, i, np.clip(gt * 0.6 + 0.3, 0, 1), gt)
    base = [os.path.join(root, "fd-gan_amd", "train.py"), "--dataroot", data, "--batchSize", "2", "--imageSize", "64", "--originalSize", "64",
            "--niter", "2", "--display", "1", "--workers", "0"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    one = subprocess.run([sys.executable] + base + ["--exp", str(tmp_path / "ck1")], env=env, capture_output=True, text=True, timeout=900)
    assert one.returncode == 0, one.stderr[-3000:]
    assert one.stdout.count(" ms/step") == 8, one.stdout[-2000:]                  # 2 epochs x ceil(7 / 2) steps, ragged last batch included
    assert os.path.exists(str(tmp_path / "ck1" / "netG_epoch_1.pth"))
    env2 = dict(env, FDGAN_DP_BACKEND="gloo", FDGAN_DP_SHARED_GPU="1")
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(_free_port())] + base + ["--exp", str(tmp_path / "ck2")], env=env2, capture_output=True, text=True,
                         timeout=900)
    assert two.returncode == 0, two.stderr[-3000:]
    assert two.stdout.count(" ms/step") == 2, two.stdout[-2000:]                  # rank 0 prints: 2 epochs x (7 // (2 ranks x 2)) = 1 step each
    import torch
    sd = torch.load(str(tmp_path / "ck2" / "netG_epoch_1.pth"), map_location="cpu")
    assert all(k.startswith("module.") for k in sd) and len(sd) == 786
