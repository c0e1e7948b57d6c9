"""Data-parallel sharding of the hot path: one process per GPU, no data-path collective.

The reference spreads the generator over the visible GPUs with nn.DataParallel (demo.py:89):
the batch dimension is split, every replica holds the same weights and BatchNorm uses the
statistics of ITS shard.  Here that is one process per GPU (torchrun / `torch.distributed`,
backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests), each rank owning a
contiguous slice of the global batch (or every world-th dataset index for demo.py).  The forward
path exchanges nothing; the only collectives are the timing ones (barrier, MAX of elapsed time).
The gradient all-reduce of the training step belongs to the backward path and is not built yet.
"""
import os

import torch
import torch.distributed as dist


class DpContext:
    def __init__(self, rank=0, world=1, local_rank=0, device=None, owns_group=False):
        self.rank, self.world, self.local_rank, self.device, self._owns = rank, world, local_rank, device, owns_group

    @classmethod
    def from_env(cls, backend=None, device=None):
        """Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* as torchrun exports them."""
        world = int(os.environ.get("WORLD_SIZE", "1"))
        rank = int(os.environ.get("RANK", "0"))
        local = int(os.environ.get("LOCAL_RANK", "0"))
        if not 0 <= rank < world:
            raise ValueError("RANK %d outside WORLD_SIZE %d" % (rank, world))
        if device is None and torch.cuda.is_available():
            device = torch.device("cuda", local)
            torch.cuda.set_device(device)
        owns = False
        if world > 1 and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on this host driver
            if backend is None:
                backend = "nccl" if (device is not None and device.type == "cuda") else "gloo"
            dist.init_process_group(backend, rank=rank, world_size=world)
            owns = True
        return cls(rank, world, local, device, owns)

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

    def throughput(self, units_this_rank, seconds):
        """Whole-job units/s: all ranks' units over the slowest rank's time."""
        return self.sum_over_ranks(units_this_rank) / self.max_over_ranks(seconds)

    def close(self):
        if self._owns and dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()
            self._owns = False
