"""Data-parallel sharding of the hot path: one process per GPU, no data-path collective.

The reference spreads the generator over the visible GPUs with nn.DataParallel (demo.py:89):
the batch dimension is split, every replica holds the same weights and BatchNorm uses the
statistics of ITS shard.  Here that is one process per GPU (torchrun / `torch.distributed`,
backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests), each rank owning a
contiguous slice of the global batch (or every world-th dataset index for demo.py).  The forward
path exchanges nothing; its only collectives are the timing ones (barrier, MAX of elapsed time).
The training step's one exchange is the gradient all-reduce: optim.FlatAdam reduces slices of its flat
gradient buffer, for the generator overlapped with the backward walk (FlatAdam.overlap); GradBuckets below
is the generic variant for optimizers without a flat buffer.
"""
import os

# The host driver only supports dmabuf IPC: RCCL's cross-process buffer sharing needs this BEFORE the HIP runtime
# initialises (first device call), so it is set at import time, not next to init_process_group.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


class DpContext:
    def __init__(self, rank=0, world=1, local_rank=0, device=None, owns_group=False):
        self.rank, self.world, self.local_rank, self.device, self._owns = rank, world, local_rank, device, owns_group
        self.force_exchange = False         # run the gradient collectives even when world == 1 (from_env)

    @classmethod
    def from_env(cls, backend=None, device=None):
        """Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* as torchrun exports them."""
        world = int(os.environ.get("WORLD_SIZE", "1"))
        rank = int(os.environ.get("RANK", "0"))
        local = int(os.environ.get("LOCAL_RANK", "0"))
        if not 0 <= rank < world:
            raise ValueError("RANK %d outside WORLD_SIZE %d" % (rank, world))
        if device is None and torch.cuda.is_available():
            device = torch.device("cuda", 0 if os.environ.get("FDGAN_DP_SHARED_GPU") == "1" else local)      # test hook: all ranks on cuda:0
            torch.cuda.set_device(device)
        owns = False
        # FDGAN_DP_FORCE_EXCHANGE=1 (tests, single-GPU boxes): build the process group and run the gradient collectives even
        # with ONE rank, so that `torchrun --nproc-per-node 1 bench.py --gpus 1` walks the whole RCCL code path
        force = os.environ.get("FDGAN_DP_FORCE_EXCHANGE") == "1"
        if (world > 1 or force) and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            if backend is None:      # FDGAN_DP_BACKEND: tests put two ranks on ONE GPU, which RCCL refuses: gloo there
                backend = os.environ.get("FDGAN_DP_BACKEND") or ("nccl" if (device is not None and device.type == "cuda") else "gloo")
            dist.init_process_group(backend, rank=rank, world_size=world)
            owns = True
        ctx = cls(rank, world, local, device, owns)
        ctx.force_exchange = force and dist.is_initialized()
        return ctx

    # ---- partitioning -------------------------------------------------------------------
    def batch_slice(self, global_batch):
        """Contiguous [start, stop) of this rank; SURVEY 8e: rank r gets samples [r*B, (r+1)*B)."""
        if global_batch % self.world:
            raise ValueError("global batch %d is not divisible by %d ranks" % (global_batch, self.world))
        per = global_batch // self.world
        return self.rank * per, (self.rank + 1) * per

    def item_indices(self, n_items):
        """Dataset indices of this rank for batch-1 inference: rank, rank+world, ... (names stay global)."""
        return list(range(self.rank, n_items, self.world))

    # ---- timing collectives -------------------------------------------------------------
    def _tensor(self, v, dtype):
        dev = self.device if (self.device is not None and dist.get_backend() == "nccl") else "cpu"
        return torch.tensor([v], dtype=dtype, device=dev)

    def barrier(self):
        if self.device is not None and self.device.type == "cuda":
            torch.cuda.synchronize()
        if self.world > 1:
            dist.barrier()
            if self.device is not None and self.device.type == "cuda":
                torch.cuda.synchronize()

    def max_over_ranks(self, seconds):
        if self.world == 1:
            return float(seconds)
        t = self._tensor(seconds, torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, count):
        if self.world == 1:
            return int(count)
        t = self._tensor(count, torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return int(t.item())

    def gather_floats(self, value):
        """[value of rank 0, ..., value of rank world-1] on every rank (diagnostics: per-rank step times)."""
        if self.world == 1:
            return [float(value)]
        t = self._tensor(0.0, torch.float64).repeat(self.world)
        t[self.rank] = float(value)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return [float(v) for v in t.tolist()]

    def throughput(self, units_this_rank, seconds):
        """Whole-job units/s: all ranks' units over the slowest rank's time."""
        return self.sum_over_ranks(units_this_rank) / self.max_over_ranks(seconds)

    def close(self):
        if self._owns and dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()
            self._owns = False


class RankBatches:
    """`batch_sampler` for a DataLoader under data parallelism: every rank runs the SAME number of equally sized steps.

    Each epoch is one permutation of the dataset drawn from an explicit generator seeded `seed + epoch` (identical on all
    ranks by construction, not by the global RNG staying in sync), cut into global batches of `world * batch` items; rank r
    takes items [r*batch, (r+1)*batch) of each -- the partitioning of DpContext.batch_slice.  The ragged tail (fewer than
    world * batch items) is DROPPED: a step that only some ranks run would issue gradient all-reduces no peer matches and
    hang the job, and a short batch on one rank would be averaged with weight 1/world.  Each rank reads only
    its own items.  The reference trains in one process (nn.DataParallel, demo.py:89) and has no such rule to mirror."""

    def __init__(self, n_items, batch, world=1, rank=0, seed=0, shuffle=True):
        if not 0 <= rank < world:
            raise ValueError("rank %d outside world %d" % (rank, world))
        self.n, self.batch, self.world, self.rank, self.seed, self.shuffle, self.epoch = n_items, batch, world, rank, seed, shuffle, 0
        if len(self) == 0:
            raise ValueError("%d items cannot fill one global batch of %d ranks x %d" % (n_items, world, batch))

    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    def __len__(self):
        return self.n // (self.batch * self.world)

    def __iter__(self):
        if self.shuffle:
            g = torch.Generator(device="cpu").manual_seed(self.seed + self.epoch)
            order = torch.randperm(self.n, generator=g).tolist()
        else:
            order = list(range(self.n))
        gb = self.batch * self.world
        for s in range(0, len(self) * gb, gb):
            yield order[s + self.rank * self.batch: s + (self.rank + 1) * self.batch]


class GradBuckets:
    """Data-parallel gradient averaging for the training path: RCCL all-reduce (backend "nccl" on the GPU
    box, "gloo" in the CPU tests) of flat fp32 buckets, launched in REVERSE parameter order so the first
    bucket (the layers whose backward finishes first) is on the wire while earlier layers are still being
    differentiated.  Only parameters that actually receive gradients are reduced -- FDGAN has 2.2 M
    parameters that never do (conv0, dense_block31, dense_norm31, the dy blocks' bn1/bn2).

    xGMI is point-to-point (7 links x ~153 GB/s per GPU): ring all-reduce time is per-link bound,
    2 (N-1)/N x bytes / link bandwidth -- 0.6 ms for the generator's 47.2 MB at N = 8 -- so a few MB per
    bucket keeps every bucket latency- rather than bandwidth-bound without fragmenting the reduction.
    """

    def __init__(self, params, ctx, bucket_mb=8.0):
        self.ctx = ctx
        self.params = [p for p in params if p.requires_grad]
        self.bucket_elems = int(bucket_mb * (1 << 20) / 4)

    def _buckets(self):
        cur, size = [], 0
        for p in reversed(self.params):
            if p.grad is None:
                continue
            cur.append(p)
            size += p.numel()
            if size >= self.bucket_elems:
                yield cur
                cur, size = [], 0
        if cur:
            yield cur

    def allreduce_(self, async_op=True):
        """Averages .grad over the ranks in place.  Returns the number of buckets reduced."""
        if self.ctx.world == 1:
            return 0
        pending = []
        for bucket in self._buckets():
            flat = torch.cat([p.grad.reshape(-1) for p in bucket])
            work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=async_op)
            pending.append((bucket, flat, work))
        for bucket, flat, work in pending:
            if work is not None and async_op:
                work.wait()
            flat.div_(self.ctx.world)
            off = 0
            for p in bucket:
                n = p.numel()
                p.grad.copy_(flat[off:off + n].view_as(p.grad))
                off += n
        return len(pending)
