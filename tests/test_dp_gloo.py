"""N>1 path on CPU: two processes over gloo exercise the sharding and the timing collectives
bench.py and demo.py use (the forward path itself has no data-path collective)."""
import os
import socket

import torch
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from fdgan_hip.dp import DpContext
    dp = DpContext.from_env(backend="gloo", device=torch.device("cpu"))
    lo, hi = dp.batch_slice(32)
    items = dp.item_indices(7)
    dp.barrier()
    slow = dp.max_over_ranks(1.0 + rank)                 # rank 1 is the slow one
    total = dp.sum_over_ranks(hi - lo)
    rate = dp.throughput(16 * 10, 2.0 * (1 + rank))
    torch.save(dict(lo=lo, hi=hi, items=items, slow=slow, total=total, rate=rate), os.path.join(out_dir, "r%d.pt" % rank))
    dp.close()


def test_two_rank_sharding_and_timing(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(os.path.join(str(tmp_path), "r%d.pt" % i)) for i in range(world)]
    assert (r[0]["lo"], r[0]["hi"], r[1]["lo"], r[1]["hi"]) == (0, 16, 16, 32)       # SURVEY 8e partitioning
    assert r[0]["items"] == [0, 2, 4, 6] and r[1]["items"] == [1, 3, 5]
    assert sorted(r[0]["items"] + r[1]["items"]) == list(range(7))
    for x in r:
        assert x["slow"] == 2.0 and x["total"] == 32
        assert x["rate"] == 2 * 16 * 10 / 4.0                                          # all units / slowest rank


def test_single_process_context_is_a_no_op(monkeypatch):
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    from fdgan_hip.dp import DpContext
    dp = DpContext.from_env(device=torch.device("cpu"))
    assert (dp.rank, dp.world) == (0, 1) and dp.batch_slice(16) == (0, 16) and dp.item_indices(3) == [0, 1, 2]
    assert dp.max_over_ranks(0.5) == 0.5 and dp.throughput(10, 2.0) == 5.0
    dp.barrier(), dp.close()
    import pytest
    monkeypatch.setenv("WORLD_SIZE", "2"), monkeypatch.setenv("RANK", "5")
    with pytest.raises(ValueError):
        DpContext.from_env(device=torch.device("cpu"))


def _grad_worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from fdgan_hip.dp import DpContext, GradBuckets
    dp = DpContext.from_env(backend="gloo", device=torch.device("cpu"))
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3), torch.nn.BatchNorm2d(8), torch.nn.Conv2d(8, 4, 1))
    frozen = torch.nn.Linear(4, 4)                       # never receives a gradient: must be skipped, not reduced
    params = list(net.parameters()) + list(frozen.parameters())
    x = torch.full((2, 3, 6, 6), float(rank + 1))
    net(x).square().mean().backward()
    local = [p.grad.clone() for p in net.parameters()]
    nb = GradBuckets(params, dp, bucket_mb=0.0002).allreduce_()
    torch.save(dict(local=local, avg=[p.grad.clone() for p in net.parameters()], nb=nb,
                    frozen_none=all(p.grad is None for p in frozen.parameters())), os.path.join(out_dir, "g%d.pt" % rank))
    dp.close()


def test_two_rank_gradient_allreduce(tmp_path):
    world = 2
    mp.spawn(_grad_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(os.path.join(str(tmp_path), "g%d.pt" % i)) for i in range(world)]
    assert r[0]["nb"] == r[1]["nb"] >= 2 and r[0]["frozen_none"] and r[1]["frozen_none"]
    for k in range(len(r[0]["local"])):
        want = (r[0]["local"][k] + r[1]["local"][k]) / 2
        assert torch.allclose(r[0]["avg"][k], want, rtol=1e-6, atol=1e-8)
        assert torch.equal(r[0]["avg"][k], r[1]["avg"][k])


def _flat_worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from fdgan_hip.dp import DpContext
    from fdgan_hip.optim import FlatAdam
    dp = DpContext.from_env(backend="gloo", device=torch.device("cpu"))
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.randn(7, 3)), torch.nn.Parameter(torch.randn(5)), torch.nn.Parameter(torch.randn(2, 2, 3, 3))]
    opt = FlatAdam(params, lr=1e-3)
    opt.zero_grad()
    sum((p * float(rank + 1)).sum() for p in params).backward()          # d/dp = rank + 1 everywhere
    nb = opt.allreduce_grads(dp, bucket_mb=0.0001)
    ok_alias = all(p.grad.data_ptr() == opt.grad.data_ptr() + 4 * o for p, o in zip(params, opt.offsets))
    try:
        opt.step()
        stepped = True
    except RuntimeError as e:
        stepped = "no CPU fallback" not in str(e)
    torch.save(dict(g=[p.grad.clone() for p in params], nb=nb, ok_alias=ok_alias, stepped=stepped), os.path.join(out_dir, "f%d.pt" % rank))
    dp.close()


def test_two_rank_flat_gradient_allreduce(tmp_path):
    world = 2
    mp.spawn(_flat_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(os.path.join(str(tmp_path), "f%d.pt" % i)) for i in range(world)]
    for x in r:
        assert x["nb"] >= 2 and x["ok_alias"] and x["stepped"] is False      # the Adam update itself needs the GPU
        for g in x["g"]:
            assert torch.allclose(g, torch.full_like(g, 1.5))                  # mean of 1 and 2


class _FakeWalk:
    """Stands in for fdgan_hip.backward.PlanBackward: records -> the parameters their backward adds to."""

    def __init__(self, per_record):
        self.recs = per_record

    def record_params(self, i):
        return list(self.recs[i])


def _overlap_worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from fdgan_hip import backward as BW
    from fdgan_hip.dp import DpContext
    from fdgan_hip.optim import FlatAdam
    dp = DpContext.from_env(backend="gloo", device=torch.device("cpu"))
    torch.manual_seed(0)
    sizes = [300, 5, 700, 64, 1200, 9, 800, 33]
    params = [torch.nn.Parameter(torch.randn(n)) for n in sizes]
    opt = FlatAdam(params)
    # forward order of use: p6 (registered late, used first), p0, p1+p2, p3, p4+p5, p7; p1 is used twice
    walk = _FakeWalk([[params[6]], [params[0]], [params[1], params[2]], [params[3]], [params[4], params[5], params[1]], [params[7]]])
    opt.zero_grad()
    with opt.overlap(dp, bucket_mb=1000 * 4 / (1 << 20)) as ov:                  # bucket = 1000 floats
        for i in range(len(walk.recs) - 1, -1, -1):                               # the reverse walk
            for p in walk.recs[i]:
                p.grad.add_(float(rank + 1) * (i + 1))                            # this record's contribution on this rank
            BW.PROGRESS_HOOK(walk, i)
        early = ov.sent_early
    assert BW.PROGRESS_HOOK is None
    torch.save(dict(grad=opt.grad.clone(), early=early), os.path.join(out_dir, "ov%d.pt" % rank))
    dp.close()


def test_two_rank_overlapped_allreduce_of_the_flat_gradient(tmp_path):
    """FlatAdam.overlap: slices go out during the walk once every contributing record has run (a parameter used by two
    records, one registered late but used first), the rest at the end; both ranks end with the mean."""
    world = 2
    mp.spawn(_overlap_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(os.path.join(str(tmp_path), "ov%d.pt" % i)) for i in range(world)]
    assert torch.equal(r[0]["grad"], r[1]["grad"])
    # expected mean over ranks of (rank + 1) * sum over records using the parameter of (record index + 1)
    uses = {0: [2], 1: [3, 5], 2: [3], 3: [4], 4: [5], 5: [5], 6: [1], 7: [6]}
    sizes = [300, 5, 700, 64, 1200, 9, 800, 33]
    off = 0
    for k, n in enumerate(sizes):
        want = 1.5 * sum(uses[k])
        assert torch.allclose(r[0]["grad"][off:off + n], torch.full((n,), want)), k
        off += (n + 3) // 4 * 4
    assert r[0]["early"] >= 1            # at least one slice left before the walk ended


def _loader_worker(rank, world, port, out_dir, root):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch.distributed as dist
    from datasets.pix2pix import pix2pix
    from fdgan_hip.dp import DpContext, RankBatches
    dp = DpContext.from_env(backend="gloo", device=torch.device("cpu"))
    ds = pix2pix(root)
    sampler = RankBatches(len(ds), 2, dp.world, dp.rank, seed=1234)
    loader = torch.utils.data.DataLoader(ds, batch_sampler=sampler, num_workers=0)
    seen, steps = [], 0
    for epoch in range(2):
        sampler.set_epoch(epoch)
        ids = []
        for haze, gt in loader:
            assert haze.shape == (2, 3, 8, 8)
            # what TrainStep.step does once per step: a collective every rank must match (the job hangs here if step counts differ)
            t = torch.tensor([float(haze[:, 0, 0, 0].sum())], dtype=torch.float64)
            dist.all_reduce(t)
            ids += [int(round(float(v) * 100)) for v in haze[:, 0, 0, 0]]
            steps += 1
        seen.append(ids)
    torch.save(dict(seen=seen, steps=steps, n=len(sampler)), os.path.join(out_dir, "l%d.pt" % rank))
    dp.close()


def test_two_rank_epoch_loader_equal_steps_with_odd_batch_count(tmp_path):
    """train.py sharded `if i % world != rank: continue` over a loader without drop_last, so with an odd number of
    batches one rank ran a step (and its all-reduces) no peer matched.  11 items, batch 2, 2 ranks = 5.5 batches: both ranks must
    run exactly 2 steps per epoch on disjoint items, the same permutation on both, a different one per epoch, tail dropped."""
    import numpy as np
    from datasets.pix2pix import write_pair
    root = str(tmp_path / "data")
    for i in range(11):
        img = np.full((8, 8, 3), i / 100.0, np.float32)
        write_pair(root, i, img, img)
    world = 2
    mp.spawn(_loader_worker, args=(world, _free_port(), str(tmp_path), root), nprocs=world, join=True)
    r = [torch.load(os.path.join(str(tmp_path), "l%d.pt" % i)) for i in range(world)]
    assert r[0]["n"] == r[1]["n"] == 2 and r[0]["steps"] == r[1]["steps"] == 4
    for e in range(2):
        a, b = r[0]["seen"][e], r[1]["seen"][e]
        assert len(a) == len(b) == 4 and not set(a) & set(b) and set(a) | set(b) <= set(range(11))
    assert r[0]["seen"][0] != r[0]["seen"][1]                        # reshuffled per epoch
    import pytest
    from fdgan_hip.dp import RankBatches
    with pytest.raises(ValueError):
        RankBatches(3, 2, 2, 0)                                      # cannot fill one global batch
    one = RankBatches(5, 2, 1, 0, shuffle=False)
    assert list(one) == [[0, 1], [2, 3]]


def test_eight_rank_rendezvous_overlap_and_loader(tmp_path):
    """configs[3] is EIGHT ranks and nothing with more than two had ever run.  The same workers as the
    two-rank tests at world size 8 over gloo: the rendezvous, the sharding / timing collectives, the overlapped slice-wise
    all-reduce of a flat gradient (mean over 8 ranks of rank + 1 = 4.5) and the epoch loader (37 items, batch 2: 2 full global
    batches of 16 per epoch, every rank 2 steps, disjoint items, ragged tail dropped)."""
    import numpy as np
    from datasets.pix2pix import write_pair
    world = 8
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(os.path.join(str(tmp_path), "r%d.pt" % i)) for i in range(world)]
    assert [(x["lo"], x["hi"]) for x in r] == [(4 * i, 4 * i + 4) for i in range(world)]
    assert sorted(sum((x["items"] for x in r), [])) == list(range(7)) and r[7]["items"] == []
    assert all(x["slow"] == 8.0 and x["total"] == 32 and x["rate"] == 8 * 16 * 10 / 16.0 for x in r)
    mp.spawn(_overlap_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(os.path.join(str(tmp_path), "ov%d.pt" % i)) for i in range(world)]
    assert all(torch.equal(r[0]["grad"], x["grad"]) for x in r[1:])
    uses = {0: [2], 1: [3, 5], 2: [3], 3: [4], 4: [5], 5: [5], 6: [1], 7: [6]}
    sizes = [300, 5, 700, 64, 1200, 9, 800, 33]
    off = 0
    for k, n in enumerate(sizes):
        assert torch.allclose(r[0]["grad"][off:off + n], torch.full((n,), 4.5 * sum(uses[k]))), k
        off += (n + 3) // 4 * 4
    assert r[0]["early"] >= 1
    root = str(tmp_path / "data")
    for i in range(37):
        img = np.full((8, 8, 3), i / 100.0, np.float32)
        write_pair(root, i, img, img)
    mp.spawn(_loader_worker, args=(world, _free_port(), str(tmp_path), root), nprocs=world, join=True)
    r = [torch.load(os.path.join(str(tmp_path), "l%d.pt" % i)) for i in range(world)]
    assert all(x["n"] == 2 and x["steps"] == 4 for x in r)
    for e in range(2):
        seen = [x["seen"][e] for x in r]
        assert all(len(s) == 4 for s in seen)
        flat = sum(seen, [])
        assert len(set(flat)) == len(flat) == 32 and set(flat) <= set(range(37))
