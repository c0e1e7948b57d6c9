"""Adam on the HIP path: the parameters that receive gradients live in ONE flat fp32 buffer (their `.data` are views
of it), so are their gradients; a step is one kernel launch over the flat tensor and the data-parallel all-reduce is
one (or a few, bucketed) collective on the flat gradient -- no per-parameter kernels, no torch.cat per step."""
import torch

from . import engine as E
from . import lib as L


class FlatAdam:
    def __init__(self, params, lr=2e-4, betas=(0.5, 0.999), eps=1e-8):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("FlatAdam: no parameter requires grad")
        dev = self.params[0].device
        sizes = [(p.numel() + 3) // 4 * 4 for p in self.params]                 # 16-byte aligned segments
        self.offsets = [0]
        for s in sizes:
            self.offsets.append(self.offsets[-1] + s)
        n = self.offsets[-1]
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self.m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.v = torch.zeros(n, dtype=torch.float32, device=dev)
        for p, o in zip(self.params, self.offsets):
            seg = self.flat[o:o + p.numel()].view_as(p)
            seg.copy_(p.data)
            p.data = seg                                                          # the parameter now aliases the flat buffer
            p.grad = self.grad[o:o + p.numel()].view_as(p)                        # and so does its gradient
            p._fd_grad_sink = p.grad                # the planned modules' backward adds into it directly (backward.grad_sink)
        self.lr, self.betas, self.eps, self.t = lr, betas, eps, 0
        # measurement aid (bench.py, N > 1): when `comm_events` is a list, every gradient exchange appends a pair of
        # events bracketing the point where the compute stream waits for the collectives = the EXPOSED communication time
        self.comm_events = None
        self.param_groups = [{"lr": lr, "params": self.params}]                  # misc.adjust_learning_rate compatibility

    def zero_grad(self, set_to_none=False):
        self.grad.zero_()
        for p, o in zip(self.params, self.offsets):                              # re-attach if someone dropped .grad
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
                p.grad = self.grad[o:o + p.numel()].view_as(p)
                p._fd_grad_sink = p.grad

    def step(self):
        E.require_gpu(self.flat, "FlatAdam.step")          # the update is a HIP kernel; no CPU fallback
        self.t += 1
        lr = self.param_groups[0]["lr"]
        L.check(L.load().fdgan_adam_step(self.flat.data_ptr(), self.grad.data_ptr(), self.m.data_ptr(), self.v.data_ptr(),
                                         self.flat.numel(), lr, self.betas[0], self.betas[1], self.eps, self.t, E.stream_ptr()),
                "adam_step")
        # the kernel wrote the parameters behind autograd's back: bump their version counters, which is what the
        # planned modules watch to re-pack their MFMA filter images (netplan.refresh_weights)
        torch.autograd.graph.increment_version(self.params)

    def overlap(self, ctx, bucket_mb=8.0, reduce_fn=None, force=False):
        """Context manager around a backward(): all-reduce slices of the flat gradient AS SOON AS every layer that
        contributes to them has run its backward, while the rest of the backward keeps computing (RCCL on its own stream;
        xGMI rings are per-link bound, so slices are >= bucket_mb to stay bandwidth- rather than latency-priced).
        Leaving the context reduces whatever is left, waits, and divides by the world size.  reduce_fn(lo, hi): test
        hook replacing the collective; force: also on a one-rank process group."""
        return _Overlap(self, ctx, bucket_mb, reduce_fn, force)

    def allreduce_grads(self, ctx, bucket_mb=8.0):
        """Data-parallel gradient averaging on the flat buffer: RCCL all-reduce of contiguous slices, last slice
        first (the parameters whose gradients are produced first by the backward walk live at the end).  Issue and
        wait in one call; allreduce_begin / allreduce_end split them so that other work can be enqueued in between."""
        return self.allreduce_end(self.allreduce_begin(ctx, bucket_mb))

    def allreduce_begin(self, ctx, bucket_mb=8.0):
        """Enqueue the collectives (RCCL's stream waits for what the CURRENT stream has enqueued so far) and return a
        handle for allreduce_end.  None on a single rank."""
        if ctx is None or (ctx.world == 1 and not getattr(ctx, "force_exchange", False)):
            return None
        import torch.distributed as dist
        n, step = self.grad.numel(), max(1, int(bucket_mb * (1 << 20) / 4))
        works = []
        for hi in range(n, 0, -step):
            lo = max(0, hi - step)
            works.append(dist.all_reduce(self.grad[lo:hi], op=dist.ReduceOp.SUM, async_op=True))
        return works, ctx.world

    def allreduce_end(self, handle):
        """Make the CURRENT stream wait for the collectives of `handle`, then divide by the world size (on that stream)."""
        if handle is None:
            return 0
        works, world = handle
        ev = _events(self)
        for w in works:
            w.wait()
        _events_done(self, ev)
        self.grad.div_(world)
        return len(works)


def _events(opt):
    if opt.comm_events is None or not opt.grad.is_cuda:
        return None
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    return e0, e1


def _events_done(opt, ev):
    if ev is not None:
        ev[1].record()
        opt.comm_events.append(ev)


class _Overlap:
    def __init__(self, opt, ctx, bucket_mb, reduce_fn, force=False):
        self.opt, self.ctx, self.reduce_fn = opt, ctx, reduce_fn
        # force: run the collectives even on a one-rank group (exercises the RCCL path on a single-GPU box)
        self.active = reduce_fn is not None or (ctx is not None and (ctx.world > 1 or force or getattr(ctx, "force_exchange", False)))
        self.bucket = max(1, int(bucket_mb * (1 << 20) / 4))
        self.index = {id(p): k for k, p in enumerate(opt.params)}
        self.tables = {}                    # id(PlanBackward) -> per-record parameter indices
        self.pending = None                 # per parameter: records of the walks seen so far that still have to run
        self.sent = [False] * len(opt.params)
        self.touched = [False] * len(opt.params)
        self.works, self.sent_early = [], 0
        self.walks = []                     # the PlanBackward objects seen (their side streams are joined before a send)

    def __enter__(self):
        if self.active:
            from . import backward as BW
            self._prev, BW.PROGRESS_HOOK = BW.PROGRESS_HOOK, self._progress
        return self

    def _progress(self, B, i):
        tab = self.tables.get(id(B))
        if tab is None:                     # first record of this walk: count every record's contributions
            self.walks.append(B)
            tab = self.tables[id(B)] = [[self.index[id(p)] for p in B.record_params(j) if id(p) in self.index]
                                        for j in range(B.num_progress_records() if hasattr(B, "num_progress_records") else len(B.recs))]
            if self.pending is None:
                self.pending = [0] * len(self.opt.params)
            for idxs in tab:
                for k in idxs:
                    self.pending[k] += 1
                    self.touched[k] = True
        for k in tab[i]:
            self.pending[k] -= 1
        self._sweep(final=False)

    def _send(self, k0, k1):
        for B in self.walks:                # weight gradients a walk put on its side stream must have landed
            if hasattr(B, "join_side"):
                B.join_side()
        lo, hi = self.opt.offsets[k0], self.opt.offsets[k1]
        for k in range(k0, k1):
            self.sent[k] = True
        if self.reduce_fn is not None:
            self.reduce_fn(lo, hi)
        else:
            import torch.distributed as dist
            self.works.append(dist.all_reduce(self.opt.grad[lo:hi], op=dist.ReduceOp.SUM, async_op=True))

    def _sweep(self, final):
        """Contiguous runs of complete, not yet reduced parameters: reduce those of at least one bucket (all, at the end)."""
        n, k = len(self.sent), 0
        while k < n:
            ready = lambda j: not self.sent[j] and (final or (self.touched[j] and self.pending[j] == 0))
            if not ready(k):
                k += 1
                continue
            k0 = k
            while k < n and ready(k):
                k += 1
            if final or self.opt.offsets[k] - self.opt.offsets[k0] >= self.bucket:
                self._send(k0, k)
                if not final:
                    self.sent_early += 1

    def __exit__(self, et, ev, tb):
        if not self.active:
            return False
        from . import backward as BW
        BW.PROGRESS_HOOK = self._prev
        if et is None:
            self._sweep(final=True)
            ev = _events(self.opt)         # the backward's kernels are all enqueued: what the stream waits for from here is exposed
            for w in self.works:
                w.wait()
            _events_done(self.opt, ev)
            if self.reduce_fn is None:
                self.opt.grad.div_(self.ctx.world)
        return False
