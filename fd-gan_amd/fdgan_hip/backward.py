"""Reverse-mode execution of a recorded NetPlan (train-mode BatchNorm), on the HIP backward entry points.

Forward ops are fused convolutions on channel-slice views of shared buffers (torch.cat never exists), so the
backward is not an autograd graph over tensors but a walk over the plan's op records in reverse:

    y = act_e(conv(pool?(act_p(bn?(x))), W) + b) [nearest x2]

      dy    = G[y]                      (2x2-summed for an upsampled output, masked for a ReLU epilogue)
      dW,db += wgrad(x, prologue, dy)   (`a` recomputed from the raw x and the forward batch statistics)
      da    = conv(dy, flip(W))         (the data gradient is a forward convolution, pad' = k-1-pad)
      G[x] += prologue_backward(da)     (un-pool, activation mask, BatchNorm backward with its two reductions)

G[.] is a gradient buffer mirroring every activation buffer (same NHWC bf16 layout, same views), zeroed
per backward; a tensor with several consumers -- the prefix of a dense block's concat buffer is read by
every later layer -- simply accumulates.  Buffers that the forward overwrites (the shared 128-channel
bottleneck of a dense block, reused by every layer) are recomputed from their producer right before the
consumer's backward needs them: one extra 1x1 forward per dense layer instead of 58 kept activations.

Reference semantics: torch.autograd over /root/reference/models/dehaze1113.py:703-801 (FDGAN), :256-275,
:358-370 (dy blocks).
"""
import contextlib
import ctypes as C
import os

import torch

from . import engine as E
from . import lib as L


class _Tape:
    """One reverse walk as it was RECORDED (round 4): multi-stream FdPlans -- every launch of the walk with its stream slot (0: the
    walk's stream, 1: the weight-gradient side stream) and the fork / join dependencies between the two -- interleaved with the few
    host callables a walk still needs (the optimizer's all-reduce hook under data parallelism, eval-mode coefficient save / restore).
    Replaying it costs one library call per plan segment instead of ~30 Python calls per record."""

    def __init__(self):
        self.lib = L.load()
        self.steps = []          # E.Plan | callable
        self.keep = []           # temporaries the recorded launches point into
        self.cur = None
        self.inplace = []        # parameters whose gradient the walk adds straight into their optimizer sink
        self.sinks = []          # (parameter, data_ptr of the sink) the recorded launches write to
        self.side_used = False
        self.cur_side = False    # the open plan segment has launches on slot 1
        self.launches = 0

    def _plan(self):
        if self.cur is None:
            self.cur = E.Plan()
            L.check(self.lib.fdgan_plan_begin(self.cur.h), "plan_begin")
        return self.cur

    def _close(self):
        if self.cur is not None:
            L.check(self.lib.fdgan_plan_end(self.cur.h), "plan_end")
            if len(self.cur):
                self.launches += len(self.cur)
                self.cur.side_used = self.cur_side
                self.steps.append(self.cur)
            self.cur = None
            self.cur_side = False

    def set_slot(self, slot):
        L.check(self.lib.fdgan_plan_set_slot(self._plan().h, slot), "plan_set_slot")
        if slot:
            self.side_used = self.cur_side = True

    def wait(self, waiter, signaler):
        L.check(self.lib.fdgan_plan_record_wait(self._plan().h, waiter, signaler), "plan_record_wait")

    def add_py(self, fn):
        self._close()
        self.steps.append(fn)
        self._plan()             # whatever the walk launches next is recorded again (launches only record while a plan is open)

    def abort(self):
        if self.cur is not None:
            self.lib.fdgan_plan_end(self.cur.h)
            self.cur = None

    def close(self):
        """Destroys the recorded plans (raw pointers into `keep` and the owner's buffers) and lets the temporaries go.  The
        owner (PlanBackward.close) has joined and drained the side stream first."""
        self.abort()
        for st in self.steps:
            if isinstance(st, E.Plan):
                st.close()
        self.steps, self.keep, self.inplace, self.sinks = [], [], [], []

    def replay(self, streams, owner=None):
        """owner: the PlanBackward whose walk this is.  Its `_w_pending` flag is raised after EVERY segment that launched on the
        side stream: a host step between two segments -- the optimizer's all-reduce hook -- calls
        `owner.join_side()`, which waits for the side stream only while the flag is up and lowers it; raised once per walk, the
        second and later buckets of a data-parallel step would be handed to RCCL without waiting for the weight gradients the
        segments since the first join put on the side stream."""
        lib = self.lib
        for st in self.steps:
            if isinstance(st, E.Plan):
                L.check(lib.fdgan_plan_launch_multi(st.h, streams, 2), "plan_launch_multi")
                if owner is not None and getattr(st, "side_used", False):
                    owner._w_pending = True
            else:
                st()



def _r8(v):
    return (v + 7) // 8 * 8


def _region(view):
    return view.buf.data_ptr(), view.c0, view.c0 + view.c


def _overlap(a, b):
    return a[0] == b[0] and a[1] < b[2] and b[1] < a[2]


class _InPlace:
    """Marker in a `grads` dict: the gradient was accumulated straight into the parameter's own `.grad` (a view of an
    optimizer's flat gradient buffer, see optim.FlatAdam); the autograd bridge returns None for it."""

    def __repr__(self):
        return "IN_PLACE"


IN_PLACE = _InPlace()

# Called as PROGRESS_HOOK(plan_backward, record_index) after each record of a reverse walk is done: the optimizer's
# gradient all-reduce rides on it (optim.FlatAdam.overlap), so RCCL runs while the rest of the backward computes.
FORCE_EAGER = False         # measurement aid (bench.py): walk eagerly even where a tape exists, so that wrapped engine entry points see every call
TR_DEFER_MAX = 1 << 24      # floats (64 MiB) of partial sums a record may keep private for the batched reduce
PROGRESS_HOOK = None


def _constant_entries(meta, cin):
    """Channel ranges of a BatchNorm prologue whose (mean, var) are CONSTANTS rather than statistics of the batch: the whole norm in
    eval mode (running statistics), the `identity` entries of a side-by-side table (models/dehaze22.py).  There the backward is
    dx = gamma * rstd * dpre: dgamma / dbeta are formed as always (the norm's own parameters still learn), the two correction
    terms -dbeta / M - xhat * dgamma / M, which differentiate the statistics, do not exist."""
    return meta.get("identity", ()) if meta.get("batch_stats", False) else ((0, cin),)


def grad_sink(p):
    """The fp32 tensor gradients of `p` may be added into directly, or None: FlatAdam marks its parameters with
    `_fd_grad_sink` (= their `.grad` view); anything else goes through autograd's own accumulation."""
    sink = getattr(p, "_fd_grad_sink", None)
    if sink is not None and p.grad is not None and p.grad.data_ptr() == sink.data_ptr():
        return sink
    return None


def grad_target(grads, p):
    """Tensor to accumulate the gradient of `p` into during this backward walk (registered in `grads`)."""
    t = grads.get(p)
    if t is IN_PLACE:
        return p._fd_grad_sink
    if t is None:
        sink = grad_sink(p)
        if sink is not None:
            grads[p] = IN_PLACE
            return sink
        t = grads[p] = torch.zeros_like(p)
    return t


def autograd_grads(grads, params):
    """What an autograd.Function returns for `params`: None where the gradient already sits in `.grad`."""
    return tuple(None if grads.get(p) is IN_PLACE else grads.get(p) for p in params)


class PlanBackward:
    def __init__(self, plan):
        self.plan = plan
        self.recs = plan.records
        self.gbuf = {}          # activation buffer data_ptr -> gradient buffer
        self.multi_version = set()
        alias = getattr(plan, "grad_alias", {})
        for r in self.recs:
            for key in ("x", "y", "src", "dst"):
                v = r.get(key)
                if v is None or v.buf.data_ptr() in self.gbuf:
                    continue
                ptr = v.buf.data_ptr()
                if ptr in alias:          # per-layer activation buffers whose gradients are consumed one at a time
                    if alias[ptr] not in self.gbuf:
                        self.gbuf[alias[ptr]] = E.grad_like(v.buf)
                    self.gbuf[ptr] = self.gbuf[alias[ptr]]
                    self.multi_version.update((ptr, alias[ptr]))
                else:
                    self.gbuf[ptr] = E.grad_like(v.buf)
        # recomputation: op i's input was produced by op j and overwritten afterwards
        self.recompute = {}
        for i, r in enumerate(self.recs):
            if r["kind"] != "conv":
                continue
            rx = _region(r["x"])
            prod = None
            for j in range(i - 1, -1, -1):
                rj = self.recs[j]
                out = rj.get("y") if rj["kind"] == "conv" else rj.get("dst")
                if out is not None and _overlap(_region(out), rx):
                    prod = j
                    break
            if prod is None:
                continue
            for k in range(i + 1, len(self.recs)):
                rk = self.recs[k]
                out = rk.get("y") if rk["kind"] == "conv" else rk.get("dst")
                if out is not None and _overlap(_region(out), rx):
                    self.recompute[i] = prod
                    self.multi_version.add(rx[0])
                    break
        dev = plan.device
        self.ws = torch.empty(1 << 26, dtype=torch.float32, device=dev)      # split-K partials (256 MiB)
        self.ws_bn = torch.empty(1 << 24, dtype=torch.float32, device=dev)   # BatchNorm-backward partial sums (64 MiB)
        self.ws_fin = torch.empty(64 * 4096, dtype=torch.float32, device=dev)  # second level of the BatchNorm-sum reduction
        self.fuse_mask = os.environ.get("FDGAN_NO_FUSED_MASK") is None        # tuning aid: separate bn_act_bwd pass
        self.defer_affine = os.environ.get("FDGAN_NO_DEFERRED_AFFINE") is None  # tuning aid: per-layer bn_bwd_apply pass
        self.fuse_wgrad = os.environ.get("FDGAN_NO_FUSED_WGRAD") is None        # tuning aid: separate 1x1 weight-gradient kernel
        self.fold_flush = os.environ.get("FDGAN_NO_FOLDED_FLUSH") is None       # tuning aid: separate affine_accumulate pass
        self.pool_one_pass = os.environ.get("FDGAN_NO_POOL_ONEPASS") is None    # tuning aid: pooled prologues through bn_bwd_apply
        # The flushed gradient of a 32-channel growth slice goes to a private PIXEL-DENSE buffer instead of back into its slice of the
        # concat-pitched gradient buffer (round 6): the slice is 64-byte pieces at a 512 .. 2048-byte pitch, and the memory system moves
        # 128-byte lines (profiles/r6_ubench_raggedrow.txt) -- its two readers, the 3x3 data- and weight-gradient kernels, then read
        # half the lines, and the flush writes whole ones.  One buffer per record (the weight gradient reads it from the side stream
        # while the walk goes on): 0.7 GB for the generator at B = 16 @ 256^2.
        self.compact_dy = os.environ.get("FDGAN_NO_COMPACT_DY") is None
        self.dyc = {}               # record index -> private dense dy buffer (n, h, w, 32)
        # Weight gradients off the critical path: dW only feeds the optimizer, while the data gradient is what the next
        # (earlier) layer waits for.  Every unfused weight gradient (kernel + its fixed-order reduction) whose operands are
        # persistent buffers runs on a second HIP stream with its own split-K workspace; the walk joins it at the end (and
        # before a gradient slice is handed to the all-reduce).  At 64 x 64 and below, where every kernel of a dense layer is
        # launch-sized, this takes two launches per layer out of the serial chain.
        self.offload_wgrad = os.environ.get("FDGAN_NO_WGRAD_STREAM") is None and dev.type == "cuda" and not self.recompute
        self.wstream = torch.cuda.Stream(device=dev, priority=int(os.environ.get("FDGAN_WGRAD_STREAM_PRIORITY", "0"))) if self.offload_wgrad else None
        self.ws_w = torch.empty(1 << 26, dtype=torch.float32, device=dev) if self.offload_wgrad else None
        self._w_pending = False
        self._zeroed = False
        self._closed = False
        # Lifetime (VERDICT r5 #2d / ADVICE r5): the gradient buffers and the side stream's workspace are allocated on the stream
        # that builds this object and read / written by kernels on `wstream`.  torch's caching allocator hands a freed block back
        # to its ALLOCATING stream at once; record_stream makes it wait for the side stream's work as well, so whenever these
        # tensors die (close(), eviction of the plan, garbage collection of the cycle plan <-> PlanBackward at a random point)
        # their memory cannot be re-used under a side-stream kernel that still points into it.
        self._share(self.ws_w, *{id(g): g for g in self.gbuf.values()}.values())
        # The fused bottleneck kernel's weight-gradient partials ([pixel slots][128][C] per dense layer) are summed by ONE
        # table-driven launch at the end of the walk instead of one 10 us launch per layer inside the chain of dependent
        # launches: every such record keeps its own partial buffer (0.7 GB for the generator at B = 16 @ 256^2: nothing next
        # to 288 GB).  Which records these are is learned by the first walk (it reduces per layer and notes where the fused
        # kernel ran); the set is fixed from the second walk on, so the all-reduce overlap can count on it.
        self.defer_reduce = os.environ.get("FDGAN_NO_DEFERRED_REDUCE") is None
        self._fused_seen = set()    # records on which the fused kernel launched (filled by every walk)
        self._deferred_idx = None   # frozen after the first walk
        self.wparts = {}            # record index -> (private partial buffer, nsplit of its last launch)
        self.reduce_jobs = []
        self.reduce_table = None
        # The same for the row-walking weight-gradient kernels on the side stream (the growth convs' 3x3, the discriminators' 3x3 /
        # 4x4): 256 partial blocks of 147 KB per growth conv, whose 40 MB reduction was one more launch per layer.  Records
        # whose partials fit TR_DEFER_MAX floats keep them in a private buffer; one launch on the side stream sums them all.
        self._tr_seen = {}          # record index -> floats of partials (first walk)
        self._tr_deferred = frozenset()
        self.tr_parts = {}          # record index -> private partial buffer
        self.tr_jobs = []
        self.tr_table = None
        self.walks_done = 0
        # Recording (the host needed 23.6 ms to enqueue a 28 ms step, most of it these walks).  The first two walks
        # run eagerly (they learn which records defer their reductions and allocate those buffers); the third is a DRY run that
        # records every launch into a _Tape and is followed by its first replay; from then on a walk is `tape.replay()`.  A tape
        # is keyed by everything that changes what a walk does (which parameters want gradients, whose input gradient is
        # skipped, whether an all-reduce hook rides along) and is only made when every gradient goes straight into an
        # optimizer's flat buffer (FlatAdam sinks: stable addresses) and the plan holds no legacy op with host-side arithmetic.
        self.tape_enabled = (os.environ.get("FDGAN_NO_BACKWARD_TAPE") is None and dev.type == "cuda" and not self.recompute and
                             all(r["kind"] in ("conv", "copy", "maxpool", "dropout") for r in self.recs))
        self.tapes = {}
        self._rec = None
        self._persist = {}
        self._zero_tables = {}
        self.deferred = {}      # activation buffer data_ptr -> pending per-channel (Bsum, Csum) of BatchNorm's backward
        # Sole consumers: a conv whose input region no other op reads between its producer and its next overwrite (the
        # dense layers' bottlenecks) STORES its data gradient instead of accumulating it; gradient buffers fed only by such
        # stores need no zeroing, neither at the start of a walk nor between the versions they hold.
        self.nozero = set()
        if self.fuse_mask and self.defer_affine and os.environ.get("FDGAN_NO_DX_STORE") is None:
            def reads(rk):
                return _region(rk["x"]) if rk["kind"] == "conv" else _region(rk["src"])

            def writes(rk):
                out = rk.get("y") if rk["kind"] == "conv" else rk.get("dst")
                return _region(out) if out is not None else None
            per_gbuf = {}
            for i, r in enumerate(self.recs):
                sole = False
                if r["kind"] == "conv":
                    rx = _region(r["x"])
                    prod = next((j for j in range(i - 1, -1, -1) if writes(self.recs[j]) is not None and _overlap(writes(self.recs[j]), rx)), None)
                    nxt = next((k for k in range(i + 1, len(self.recs)) if writes(self.recs[k]) is not None and _overlap(writes(self.recs[k]), rx)),
                               len(self.recs))
                    meta = r["pro"]._meta if r.get("pro") is not None else dict(pool=False)
                    sole = (prod is not None and r["stride"] == 1 and not meta["pool"] and r["x"].c0 % 8 == 0 and
                            not any(t != i and _overlap(reads(self.recs[t]), rx) for t in range(prod + 1, nxt)))
                    per_gbuf.setdefault(id(self.gbuf[rx[0]]), []).append(sole)
                else:
                    per_gbuf.setdefault(id(self.gbuf[reads(r)[0]]), []).append(False)
                r["_sole"] = sole
            self.nozero = {g for g, flags in per_gbuf.items() if all(flags)}
            # First writers: the LAST forward reader of a buffer is the first op of the walk to touch its gradient.  When that op
            # reads the WHOLE buffer through a pooled BatchNorm prologue (a dense block's transition: every channel, every pixel) its
            # one-pass backward can store instead of add -- the block's gradient buffer (0.5 GB for block 1) then needs no zeroing at
            # the start of a walk and is not read by that pass (fdgan_bn_act_bwd_dx).
            if self.pool_one_pass and self.defer_affine and os.environ.get("FDGAN_NO_FIRST_WRITER_STORE") is None:
                last_reader = {}
                for i, r in enumerate(self.recs):
                    reg = reads(r) if r["kind"] in ("conv", "copy", "maxpool") else None
                    if reg is not None:
                        last_reader[reg[0]] = i
                    elif r["kind"] not in ("dropout",):      # an op this analysis does not model: leave its buffers alone
                        for key in ("src", "dst"):
                            v = r.get(key)
                            if v is not None and hasattr(v, "buf"):
                                last_reader[v.buf.data_ptr()] = -1
                for ptr, i in last_reader.items():
                    if i < 0:
                        continue
                    r = self.recs[i]
                    if r["kind"] != "conv" or r.get("pro") is None:
                        continue
                    meta, x = r["pro"]._meta, r["x"]
                    g = self.gbuf.get(ptr)
                    whole = x.c0 == 0 and x.c == x.buf.shape[-1] and not isinstance(x, E.StridedView)
                    if (g is not None and whole and meta["pool"] and meta.get("bn") is not None and r["stride"] == 1 and r["k"] == 1
                            and ptr not in self.multi_version and sum(1 for q in self.gbuf.values() if q is g) == 1):
                        r["_first_full"] = True
                        self.nozero.add(id(g))
        # Verification hook: tests set `checks` to a list AND `check_reference` to a callable (r, dy_view, meta) -> (dW, dx)
        # that states the single fused op under torch autograd (tests/hiputil.op_reference); every op's two gradients are
        # then compared with it in place.  No reference implementation lives in the product package.
        self.checks = None
        self.check_reference = None
        # ReLU pre-masking (enable_relu_premask): whether each conv's input region is the stored output of a ReLU epilogue
        # (directly, or through max-pools / copies of one)
        self.relu_premask = False
        post, self.x_post_relu = {}, {}
        for i, r in enumerate(self.recs):
            if r["kind"] == "conv":
                self.x_post_relu[i] = post.get(_region(r["x"]), False)
                if r.get("y") is not None:
                    post[_region(r["y"])] = r["e_act"] == L.ACT_RELU
            else:
                post[_region(r["dst"])] = post.get(_region(r["src"]), False)

    def enable_relu_premask(self):
        """Fold the backward of every ReLU EPILOGUE into the op that produces the gradient of its output: a consumer's
        data-gradient epilogue masks by (stored output > 0) -- relu'(z) = [relu(z) > 0] -- so the walk needs no separate
        mask pass per layer (10 for Vgg16).  The caller seeds masked gradients (fdgan_mse_nhwc_bwd(relu_mask = 1)).  Valid
        when every reader of a ReLU output is a prologue-free stride-1 conv on the fused data-gradient path, a max-pool
        (routes to the arg-max, whose value is the pooled value: masked with it) or a copy."""
        if self.relu_premask:
            return
        for i, r in enumerate(self.recs):
            if r["kind"] != "conv":
                continue
            meta = r["pro"]._meta if r.get("pro") is not None else None
            if self.x_post_relu[i] and not (r["stride"] == 1 and meta is None and r["x"].c0 % 8 == 0 and not r["upsample"]):
                raise NotImplementedError("ReLU pre-masking: record %d reads a ReLU output through a prologue / stride" % i)
        if not (self.fuse_mask and self.defer_affine):
            raise NotImplementedError("ReLU pre-masking needs the fused data-gradient path")
        self.relu_premask = True

    def deferred_records(self):
        """Records whose weight gradient leaves through the batched reduce at the end of the walk."""
        return (self._deferred_idx | self._tr_deferred) if self._deferred_idx is not None else frozenset()

    def num_progress_records(self):
        """PROGRESS_HOOK is called with 0 .. len(recs): the last index is the pseudo-record of the batched reduce."""
        return len(self.recs) + 1

    def record_params(self, i):
        """Parameters whose gradient record i's backward adds to (conv weight, bias, the prologue's BatchNorm pair).
        i == len(self.recs): the deferred reductions' conv weights, complete only after the batched reduce."""
        if i == len(self.recs):
            return [self.recs[j]["w"].param for j in sorted(self.deferred_records()) if self.recs[j]["w"].param.requires_grad]
        r = self.recs[i]
        if r["kind"] != "conv":
            return []
        out = [] if i in self.deferred_records() else [r["w"].param]
        if r.get("bias") is not None:
            out.append(r["bias"])
        bn = r["pro"]._meta.get("bn") if r.get("pro") is not None else None
        if bn is not None and bn.weight is not None:
            out += [bn.weight, bn.bias]
        return [p for p in out if p.requires_grad]

    def _share(self, *tensors):
        """Tensors the weight-gradient side stream reads or writes, allocated on another stream (see __init__)."""
        if self.wstream is not None:
            for t in tensors:
                if t is not None and t.is_cuda:
                    t.record_stream(self.wstream)

    def close(self):
        """Deterministic teardown: the side stream is joined and drained BEFORE any buffer its launches point into can be freed,
        recorded tapes (raw pointers into those buffers) are destroyed, and the reference cycle plan <-> PlanBackward is broken
        so that the buffers die here and not at a random garbage collection inside later work.  Idempotent; called by
        NetPlan.close() (plan eviction) and by __del__."""
        if getattr(self, "_closed", True):
            return
        self._closed = True
        dev = self.plan.device if self.plan is not None else None
        if self.wstream is not None and dev is not None and dev.type == "cuda":
            try:
                torch.cuda.current_stream(dev).wait_stream(self.wstream)
                self.wstream.synchronize()
            except Exception:
                pass                                  # interpreter shutdown: the device context may be gone
        self._w_pending = False
        for tape in self.tapes.values():
            if tape is not None:
                tape.close()
        self.tapes.clear()
        self._rec = None
        for name in ("gbuf", "wparts", "tr_parts", "dyc", "_persist", "_zero_tables", "deferred"):
            getattr(self, name).clear()
        self.ws = self.ws_bn = self.ws_fin = self.ws_w = None
        self.reduce_table = self.tr_table = None
        self.reduce_jobs, self.tr_jobs = [], []
        self.plan = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def join_side(self):
        """The walk's stream waits for the weight gradients issued on the side stream so far (end of a walk; before a slice
        of the flat gradient is handed to the all-reduce)."""
        if self._w_pending:
            if self._rec is not None:
                self._rec.wait(0, 1)
            else:
                torch.cuda.current_stream(self.wstream.device).wait_stream(self.wstream)
            self._w_pending = False

    # ---- eager / recorded execution of everything that is not a plain library launch --------------------------------
    def _py(self, fn):
        """A host-side step of the walk: run now, or (recording) become a step of the tape."""
        if self._rec is None:
            fn()
        else:
            self._rec.add_py(fn)

    def _tmp(self, shape, dtype):
        """A temporary of this walk; a recorded walk keeps it (the launches point into it)."""
        t = torch.empty(shape, dtype=dtype, device=self.plan.device)
        if self._rec is not None:
            self._rec.keep.append(t)
            self._share(t)                           # a tape's launches may read it from the side stream for as long as the tape lives
        return t

    def _new_grad(self, n, h, w, c, zero=False):
        t = self._tmp((n, h, w, c), E.GRAD_DTYPE)
        if zero:
            E.fill_zero(t)
        return t

    @contextlib.contextmanager
    def _side(self):
        """Launches inside go to the weight-gradient side stream (recording: to stream slot 1)."""
        if self._rec is None:
            with torch.cuda.stream(self.wstream):
                yield
        else:
            self._rec.set_slot(1)
            try:
                yield
            finally:
                self._rec.set_slot(0)

    def _fork(self):
        """The side stream waits for what the walk's stream has enqueued so far."""
        if self._rec is None:
            self.wstream.wait_stream(torch.cuda.current_stream(self.plan.device))
        else:
            self._rec.wait(1, 0)

    def persistent(self, name, make):
        """A buffer that lives as long as this object (the seed gradient of a network's last conv: recorded launches read it)."""
        t = self._persist.get(name)
        if t is None:
            t = self._persist[name] = make()
        return t

    def G(self, view):
        g = self.gbuf[view.buf.data_ptr()]
        if isinstance(view, E.StridedView):      # one parity of a transposed conv's output: the same pixels of the gradient buffer
            return E.StridedView(g, view.c0, view.c, *view.geom)
        return E.View(g, view.c0, view.c)

    def zero_(self):
        """Every gradient buffer that is accumulated into: ONE launch over a device table (they were ~25 torch fills per walk)."""
        skip = self.nozero if self.checks is None else ()
        tab = self._zero_tables.get(self.checks is None)
        if tab is None:
            bufs = [g for g in {id(g): g for g in self.gbuf.values()}.values() if id(g) not in skip]      # aliased buffers once
            bufs += [d["coef"] for d in self.deferred.values()]      # the pending BatchNorm coefficient pairs: clean at the start of every walk
            if all((g.numel() * g.element_size()) % 16 == 0 and g.data_ptr() % 16 == 0 for g in bufs):
                tab = E.ZeroTable(bufs) if bufs else False
            else:
                tab = bufs
            self._zero_tables[self.checks is None] = tab
        if isinstance(tab, list):
            for g in tab:
                g.zero_()
        elif tab:
            tab.launch()
        for d in self.deferred.values():                                # the launch above cleared every pair
            d["dirty"].clear()
            d["stale"].clear()
        self._zeroed = True

    # ---- one fused convolution ---------------------------------------------------------------
    def _fuse_w(self, r, need_dx):
        """The dense-layer bottleneck (1x1, 128 filters): data and weight gradient in one pass over dy and x."""
        x, w, pro = r["x"], r["w"], r["pro"]
        pool = pro._meta["pool"] if pro is not None else False
        return (self.fuse_wgrad and w.param.requires_grad and need_dx and self.checks is None and r["k"] == 1 and r["stride"] == 1 and
                not w.transposed and r["bias"] is None and w.cout == 128 and r["y"] is not None and r["y"].c == 128 and not pool and
                x.c0 % 8 == 0 and self.fuse_mask and self.defer_affine)

    def _pending_for_fused(self, r, need_dx):
        """The deferred coefficient record of r's output buffer when the fused bottleneck kernel can apply it itself (the
        whole 128-channel buffer is r's output and nothing else reads its gradient): no flush pass then."""
        if not (self.fold_flush and self._fuse_w(r, need_dx)) or r["upsample"] or r["e_act"] != L.ACT_NONE:
            return None
        y = r["y"]
        d = self.deferred.get(y.buf.data_ptr())
        if d is None or not d["dirty"] or y.c0 != 0 or d["buf"].shape[-1] != 128:
            return None
        return d

    def conv_backward(self, r, dy_view, grads, need_dx=True, dy_pending=None):
        """dy_view: gradient of the op's stored output (already sum-pooled / masked by the caller if needed).  dy_pending:
        _pending_for_fused's record -- dy still lacks B * y + C, which the fused kernel adds on the fly."""
        x, w, k, pad, pro = r["x"], r["w"], r["k"], r["pad"], r["pro"]
        meta = pro._meta if pro is not None else dict(act=L.ACT_NONE, pool=False, bn=None)
        if r["stride"] != 1 and need_dx and (w.transposed or (pro is not None and pro._meta["pool"])):
            raise NotImplementedError("data gradient of a strided transposed / pooled conv")
        desc = E.conv_desc(k, r["stride"], pad, cout=w.cout)
        # ---- parameters
        p = w.param
        fuse_w = self._fuse_w(r, need_dx) and dy_view.c == 128      # fused below
        assert dy_pending is None or fuse_w
        if p.requires_grad and not fuse_w:
            if w.transposed:                      # ConvTranspose2d 1x1: weight is (cin, cout, 1, 1)
                tmp = self._tmp((w.cout, w.cin, k, k), torch.float32)
                E.conv_bwd_weight(x.fd, pro, dy_view.fd, desc, tmp, None, self.ws, False)
                tgt = grad_target(grads, p)
                if k == 1 and tgt.is_contiguous():
                    E.add_transposed(tgt, tmp)
                else:
                    self._py(lambda: tgt.add_(tmp.permute(1, 0, 2, 3)))
            else:
                db = None
                if r["bias"] is not None and r["bias"].requires_grad:
                    db = grad_target(grads, r["bias"])
                dw_t = grad_target(grads, p)
                # dy must be a view of a persistent gradient buffer that nothing rewrites before the walk ends (not the
                # dense blocks' shared bottleneck-gradient buffer, not a temporary of this call)
                side = (self.offload_wgrad and self.checks is None and r.get("y") is not None
                        and (dy_view.buf is self.gbuf.get(r["y"].buf.data_ptr()) or dy_view.buf is self.dyc.get(r.get("_idx")))
                        and r["y"].buf.data_ptr() not in self.multi_version)
                if side:
                    self._share(x.buf, dw_t, db)              # the plan's activation buffer and the gradient targets: used over there
                    self._fork()                              # dy is final (flushed) on the walk's stream
                    idx = r.get("_idx")
                    with self._side():
                        if db is not None or idx is None or not self.defer_reduce:
                            E.conv_bwd_weight(x.fd, pro, dy_view.fd, desc, dw_t, db, self.ws_w, True)
                        elif idx in self._tr_deferred:      # partial sums stay in the record's own buffer until the batched launch
                            job = E.conv_bwd_weight_job(x.fd, pro, dy_view.fd, desc, dw_t, self.tr_parts[idx], True, True)
                            assert job is not None, "a deferred weight-gradient reduction changed kernels between walks"
                            self.tr_jobs.append(job)
                        else:
                            job = E.conv_bwd_weight_job(x.fd, pro, dy_view.fd, desc, dw_t, self.ws_w, False, True)
                            if job is not None and self._deferred_idx is None:
                                self._tr_seen[idx] = job.items * job.item_stride
                    self._w_pending = True
                    if os.environ.get("FDGAN_WGRAD_SERIAL") is not None:      # debug aid: side stream, but no concurrency
                        self.join_side()
                else:
                    E.conv_bwd_weight(x.fd, pro, dy_view.fd, desc, dw_t, db, self.ws, True)
        check = self.checks is not None
        if check:
            dw_ref, dx_ref = self.check_reference(r, dy_view, meta)      # supplied by the test (tests/hiputil.op_reference)
            tmp = torch.empty((w.cout, w.cin, k, k), dtype=torch.float32, device=p.device)
            E.conv_bwd_weight(x.fd, pro, dy_view.fd, desc, tmp, None, self.ws, False)
            # the op's input-gradient contribution is measured in ISOLATION: the accumulated gradient of the region is
            # set aside and the region zeroed, so what the op adds is read back rounded to bf16 once (2^-9 relative) instead
            # of as the difference of two bf16 accumulator states (which made small contributions look wrong)
            gx_before = None
            if need_dx:
                self.flush(x)
                gslice = self.gbuf[x.buf.data_ptr()][..., x.c0:x.c0 + x.c]
                gx_before = gslice.clone()
                gslice.zero_()
            rec = dict(label="%dx%d %d->%d @%dx%d%s%s" % (k, k, w.cin, w.cout, dy_view.shape[1], dy_view.shape[2],
                                                       " pool" if meta["pool"] else "", " bn" if meta.get("bn") is not None else ""),
                       dw=float((tmp - dw_ref).norm() / (dw_ref.norm() + 1e-30)),
                       dw_hip_finite=bool(torch.isfinite(tmp).all()), dw_ref_finite=bool(torch.isfinite(dw_ref).all()))
        if not need_dx:
            if r.get("_first_full", False) and self.checks is None:      # nobody stores into this buffer now: zero it after all
                E.fill_zero(self.gbuf[x.buf.data_ptr()])
            if check:
                self.checks.append(rec)
            return
        # ---- data gradient: forward conv of dy with the flipped filter, at the conv-input resolution
        n, hy, wy, _ = dy_view.shape
        cin = w.cin
        if r["stride"] != 1:     # any-stride direct kernel (one thread per input element): the discriminators' 4x4 s2 convs
            hin, win = x.shape[1], x.shape[2]
            T = self._new_grad(n, hin, win, _r8(cin), zero=True)
            E.conv_bwd_data_direct_nhwc(dy_view.fd, p.detach().contiguous(), desc, E.View(T, 0, cin), cin)
            return self._prologue_backward(r, T, meta, grads, check_state=(rec, gx_before, dx_ref) if check else None)
        hin, win = hy + k - 1 - 2 * pad, wy + k - 1 - 2 * pad
        # the forward filter tensor in OIHW terms: ConvTranspose2d stores (cin, cout): already the transposed one
        pw = self.plan.flipped_weight(w)      # packed with the forward images, once per parameter update
        ddesc = E.conv_desc(k, 1, k - 1 - pad, cout=cin, w_layout=L.WLAYOUT_CHUNK32)
        fusable = not meta["pool"] and x.c0 % 8 == 0 and self.fuse_mask
        bn = meta.get("bn")
        if fusable:
            mask_act = meta["act"]
            if self.relu_premask and bn is None and mask_act == L.ACT_NONE and r.get("_post_relu", False):
                mask_act = L.ACT_RELU          # x is a stored ReLU output: masking by (x > 0) is that ReLU's backward
            act_pro = (E.make_prologue(act=meta["act"], mean=meta["mean"], var=meta["var"], gamma=meta["gamma"], beta=meta["beta"],
                                       eps=meta["eps"]) if bn is not None else E.make_prologue(act=mask_act))
        if fusable and self.defer_affine:
            # One pass: the data-gradient kernel masks, sums BatchNorm's two reductions and adds gamma * rstd * dpre straight
            # into G[x]; what is left of BatchNorm's backward, B * x + C per channel, is linear in x and waits in the buffer's
            # coefficient pair until the gradient of those channels is read (flush) -- shared by every layer that normalises
            # them (dense blocks).  dpre is never stored.
            gx = self.G(x)
            store = r.get("_sole", False) and self.checks is None and id(self.gbuf[x.buf.data_ptr()]) in self.nozero
            res = None
            if fuse_w:
                aff = None
                if dy_pending is not None:
                    aff = (E.View(dy_pending["buf"], 0, 128).fd, dy_pending["coef"][0], dy_pending["coef"][1])
                idx = r.get("_idx")
                if idx in self.deferred_records():      # partials into the record's own buffer, summed by the batched launch
                    ent = self.wparts.get(idx)
                    if ent is None:
                        nct = (w.cin + 127) // 128
                        ent = self.wparts[idx] = [torch.empty(max(1, 256 // nct) * 128 * w.cin, dtype=torch.float32, device=p.device), 0]
                    res = E.conv1x1_bwd_data_weight(dy_view.fd, pw, x.fd, act_pro, gx.fd, self.ws_bn if bn is not None else None,
                                                    2 if store else 1, ent[0], None, True, dy_affine=aff)
                    if res is not None:
                        self.reduce_jobs.append((ent[0], grad_target(grads, p).view(w.cout, w.cin), w.cout * w.cin, res[2], True))
                        res = res[:2]
                else:
                    res = E.conv1x1_bwd_data_weight(dy_view.fd, pw, x.fd, act_pro, gx.fd, self.ws_bn if bn is not None else None,
                                                    2 if store else 1, self.ws, grad_target(grads, p).view(w.cout, w.cin), True,
                                                    dy_affine=aff)
                    if res is not None and idx is not None:
                        self._fused_seen.add(idx)
                if res is not None and dy_pending is not None:
                    if not dy_pending.get("store"):
                        E.fill_zero(dy_pending["coef"])
                    dy_pending["dirty"].clear()
                if res is None:     # outside the fused kernel's shapes: the two separate kernels
                    if dy_pending is not None:
                        self.flush(r["y"])
                    E.conv_bwd_weight(x.fd, pro, dy_view.fd, desc, grad_target(grads, p), None, self.ws, True)
            if res is None:
                res = E.conv_bwd_data(dy_view.fd, pw, x.fd, act_pro, gx.fd, ddesc, self.ws_bn if bn is not None else None,
                                      accumulate=2 if store else 1)
            rows, cpad = res
            if bn is not None:
                train_bn = bn.weight is not None and bn.weight.requires_grad
                d = self._deferred(x)
                # table entries that are constants, not batch statistics (eval-mode BatchNorm, finished values behind a BatchNorm +
                # Dropout2d pass) get no correction terms.  finalize_coef ADDS into the buffer's shared pair, so their columns are
                # set aside and put back: zeroing them afterwards would also erase the terms other consumers of the same channels
                # (a train-mode transition behind eval-mode dense layers) added earlier in the walk and have not flushed
                keep = []
                const = [(x.c0 + lo, x.c0 + hi) for lo, hi in _constant_entries(meta, cin)]
                if const:
                    self._py(lambda: keep.extend(d["coef"][:, lo:hi].clone() for lo, hi in const))
                # a tensor this norm alone normalises, whole (a dense layer's bottleneck): its pair is WRITTEN, so nothing ever has
                # to zero it (43 fills per generator walk); everybody else's pair collects several norms and is added to
                d["store"] = store_coef = bool(r.get("_sole", False) and x.c0 == 0 and cin == d["buf"].shape[-1] and not const and self.checks is None)
                if not store_coef:
                    self._unstale(d, x.c0, x.c0 + cin)
                E.bn_bwd_finalize_coef(self.ws_bn, rows, cpad, cin, act_pro, n * hin * win, d["coef"][0, x.c0:x.c0 + cin],
                                       d["coef"][1, x.c0:x.c0 + cin],
                                       sink_dgamma=grad_target(grads, bn.weight) if train_bn else None,
                                       sink_dbeta=grad_target(grads, bn.bias) if train_bn else None, scratch=self.ws_fin, store=store_coef)
                if const:
                    def restore():
                        for (lo, hi), saved in zip(const, keep):
                            d["coef"][:, lo:hi].copy_(saved)
                        del keep[:]
                    self._py(restore)
                d["dirty"].update(range(x.c0, x.c0 + cin))
            if check:
                self.flush(x)
                self._finish_check(rec, x, gx_before, dx_ref)
            return
        T = self._new_grad(n, hin, win, _r8(cin))
        masked = None
        if fusable:     # the activation mask and the BatchNorm sums ride in the data-gradient kernel's epilogue
            masked = E.conv_bwd_data(dy_view.fd, pw, x.fd, act_pro, E.View(T, 0, cin).fd, ddesc, self.ws_bn if bn is not None else None)
        else:
            E.conv2d(dy_view.fd, pw, None, None, E.View(T).fd, ddesc)
        return self._prologue_backward(r, T, meta, grads, check_state=(rec, gx_before, dx_ref) if check else None, masked=masked)

    def _finish_check(self, rec, x, gx_before, dx_ref):
        """Verification aid: G[x] holds ONLY this op's contribution (conv_backward zeroed the region); compare it with
        torch autograd's, then put the set-aside gradient back on top."""
        added = self.G(x).torch_nchw()
        rec["dx"] = float((added - dx_ref).norm() / (dx_ref.norm() + 1e-30))
        rec["dx_scale"] = float(dx_ref.abs().mean() / (gx_before.float().abs().mean() + 1e-30))
        gslice = self.gbuf[x.buf.data_ptr()][..., x.c0:x.c0 + x.c]
        gslice.copy_((gslice.float() + gx_before.float()).to(gslice.dtype))
        self.checks.append(rec)

    # ---- deferred affine part of BatchNorm's backward (see conv_backward) ----------------------------------------
    def _deferred(self, view):
        d = self.deferred.get(view.buf.data_ptr())
        if d is None:
            d = self.deferred[view.buf.data_ptr()] = dict(
                buf=view.buf, coef=torch.zeros((2, view.buf.shape[-1]), dtype=torch.float32, device=view.buf.device), dirty=set(), stale=set(),
                store=False)
            self._zero_tables.clear()         # the pair joins the buffers zero_() clears
        return d

    def _unstale(self, d, c0, c1):
        """Channels whose pair was applied (flush) but not zeroed yet -- they are only zeroed at the start of the next walk, in
        zero_()'s one launch -- must be zero before anything is ADDED to them again in this walk (never inside a dense block:
        the slice a layer flushes lies above every prefix that is normalised afterwards)."""
        hit = sorted(c for c in d["stale"] if c0 <= c < c1)
        while hit:
            lo = hi = hit[0]
            while hit and hit[0] == hi:
                hit.pop(0)
                hi += 1
            E.fill_zero(d["coef"][:, lo:hi])
            d["stale"].difference_update(range(lo, hi))

    def flush(self, view):
        """Before G[view] is read: add the pending Bsum * x + Csum of its channels."""
        d = self.deferred.get(view.buf.data_ptr())
        if d is None or not d["dirty"]:
            return
        c0, c1 = view.c0, view.c0 + view.c
        if d["dirty"].isdisjoint(range(c0, c1)):
            return
        lo, hi = c0 - c0 % 8, min(d["buf"].shape[-1], (c1 + 7) // 8 * 8)      # whole 8-channel groups (clean ones add zero)
        for c in [c for c in range(lo, hi) if c in d["stale"] and c not in d["dirty"]]:      # applied earlier in this walk, not zeroed yet
            self._unstale(d, c, c + 1)
        xv = E.View(d["buf"], lo, hi - lo)
        E.affine_accumulate(xv.fd, d["coef"][0, lo:hi], d["coef"][1, lo:hi], self.G(xv).fd)
        if d.get("store") or self.checks is not None:
            E.fill_zero(d["coef"][:, lo:hi])          # (a written pair is never in zero_()'s table; verification walks flush out of order)
        else:
            d["stale"].update(range(lo, hi))          # zeroed by zero_() at the start of the next walk, or by _unstale before a += in this one
        d["dirty"].difference_update(range(lo, hi))

    def _flush_compact(self, i, r, y):
        """flush(y) for the output slice of a dense layer's growth conv, written to the record's private dense buffer; returns that
        buffer's view (the dy both gradient kernels of the conv read), or None when this is not such a slice / nothing is pending."""
        if not (self.compact_dy and self.checks is None and isinstance(y, E.View) and y.c == 32 and y.c0 % 32 == 0 and r["k"] == 3
                and r["stride"] == 1 and not r["upsample"] and r["e_act"] == L.ACT_NONE and r.get("bias") is None
                and y.buf.shape[-1] > 32 and y.buf.data_ptr() not in self.multi_version):
            return None
        d = self.deferred.get(y.buf.data_ptr())
        lo, hi = y.c0, y.c0 + 32
        if d is None or not d["dirty"] or d["dirty"].isdisjoint(range(lo, hi)):      # (clean channels of the slice hold a zero pair)
            return None
        for c in [c for c in range(lo, hi) if c in d["stale"] and c not in d["dirty"]]:
            self._unstale(d, c, c + 1)
        buf = self.dyc.get(i)
        if buf is None:
            n, h, w, _ = y.buf.shape
            buf = self.dyc[i] = E.new_grad(n, h, w, 32, y.buf.device)
            self._share(buf)
        xv, out = E.View(d["buf"], lo, 32), E.View(buf)
        E.affine_accumulate(xv.fd, d["coef"][0, lo:hi], d["coef"][1, lo:hi], self.G(xv).fd, out_fd=out.fd)
        if d.get("store"):
            E.fill_zero(d["coef"][:, lo:hi])
        else:
            d["stale"].update(range(lo, hi))
        d["dirty"].difference_update(range(lo, hi))
        return out

    def flush_all(self):
        for d in self.deferred.values():
            while d["dirty"]:
                c = min(d["dirty"])
                hi = c
                while hi in d["dirty"]:
                    hi += 1
                self.flush(E.View(d["buf"], c, hi - c))

    def _prologue_backward(self, r, T, meta, grads, check_state=None, masked=None):
        """G[x] += backward of (pool?, activation, BatchNorm) applied to da = T.  masked = (rows, cpad): T already holds
        dpre = da * act'(bn(x)) and ws_bn the raw-moment partials (conv_bwd_data did the first pass)."""
        x, w = r["x"], r["w"]
        p, cin = w.param, w.cin
        n, hin, win = T.shape[0], T.shape[1], T.shape[2]
        check = check_state is not None
        if check:
            rec, gx_before, dx_ref = check_state
        Tv = E.View(T, 0, cin)
        gx = self.G(x)
        bn = meta.get("bn")
        if meta["pool"] and bn is not None:
            # pooled prologue (transitions): T is the gradient w.r.t. the 2x2-averaged activation at HALF resolution.  Two
            # passes over the full-resolution input, both un-pooling and masking on the fly: the sums, then dx -- no
            # full-resolution dpre tensor (it was written, masked in place and re-read: four more passes)
            pool_pro = E.make_prologue(act=meta["act"], pool=True, mean=meta["mean"], var=meta["var"], gamma=meta["gamma"],
                                       beta=meta["beta"], eps=meta["eps"])
            dg = self._tmp((cin,), torch.float32)
            dbt = self._tmp((cin,), torch.float32)
            train_bn = bn.weight is not None and bn.weight.requires_grad
            one_pass = self.defer_affine and not check and self.pool_one_pass     # G += gamma * rstd * dpre rides in the pass that forms the sums
            # (the buffer's first writer of the walk, reading all of it: store -- its gradient buffer is not zeroed, see __init__)
            first = bool(r.get("_first_full", False)) and one_pass and self.checks is None
            assert first or not r.get("_first_full", False) or self.checks is not None, "a first-writer record left the one-pass path"
            rows, cpad = E.bn_act_bwd(Tv.fd, x.fd, pool_pro, self.ws_bn, dx_fd=gx.fd if one_pass else None, dx_store=first)
            E.bn_bwd_finalize(self.ws_bn, rows, cpad, cin, dg, dbt, sink_dgamma=grad_target(grads, bn.weight) if train_bn else None,
                              sink_dbeta=grad_target(grads, bn.bias) if train_bn else None)
            for lo, hi in _constant_entries(meta, cin):
                E.fill_zero(dg[lo:hi])
                E.fill_zero(dbt[lo:hi])
            if one_pass:      # what is left, B * x + C per channel, waits in the buffer's coefficient pair like every other norm's
                d = self._deferred(x)
                self._unstale(d, x.c0, x.c0 + cin)
                E.bn_bwd_coef(dg, dbt, pool_pro, cin, n * 4 * hin * win, d["coef"][0, x.c0:x.c0 + cin], d["coef"][1, x.c0:x.c0 + cin])
                d["dirty"].update(range(x.c0, x.c0 + cin))
                return
            E.bn_bwd_apply(Tv.fd, x.fd, pool_pro, dg, dbt, gx.fd, accumulate=True)
            if check:
                self._finish_check(rec, x, gx_before, dx_ref)
            return
        if meta["pool"]:
            T2 = self._new_grad(n, 2 * hin, 2 * win, _r8(cin))
            E.grad_ew(E.GRAD_UNPOOL, Tv, E.View(T2, 0, cin))
            T, Tv = T2, E.View(T2, 0, cin)
        if bn is not None:
            act_pro = E.make_prologue(act=meta["act"], mean=meta["mean"], var=meta["var"], gamma=meta["gamma"],
                                      beta=meta["beta"], eps=meta["eps"])
            dg = self._tmp((cin,), torch.float32)
            dbt = self._tmp((cin,), torch.float32)
            train_bn = bn.weight is not None and bn.weight.requires_grad
            sinks = dict(sink_dgamma=grad_target(grads, bn.weight) if train_bn else None,
                         sink_dbeta=grad_target(grads, bn.bias) if train_bn else None)
            if masked is not None:
                E.bn_bwd_finalize_raw(self.ws_bn, masked[0], masked[1], cin, meta["mean"], meta["var"], meta["eps"], dg, dbt,
                                      scratch=self.ws_fin, **sinks)
            else:
                rows, cpad = E.bn_act_bwd(Tv.fd, x.fd, act_pro, self.ws_bn)
                E.bn_bwd_finalize(self.ws_bn, rows, cpad, cin, dg, dbt, **sinks)
            for lo, hi in _constant_entries(meta, cin):  # constants, not batch statistics (see conv_backward): dx = A * dpre there
                E.fill_zero(dg[lo:hi])
                E.fill_zero(dbt[lo:hi])
            E.bn_bwd_apply(Tv.fd, x.fd, act_pro, dg, dbt, gx.fd, accumulate=True)
        else:
            if meta["act"] != L.ACT_NONE and masked is None:
                E.bn_act_bwd(Tv.fd, x.fd, E.make_prologue(act=meta["act"]))
            E.grad_ew(E.GRAD_ADD, Tv, gx)
        if check:
            self._finish_check(rec, x, gx_before, dx_ref)

    # ---- the legacy networks' ops ---------------------------------------------------------------
    def _legacy_backward(self, r, grads):
        src, dst = r["src"], r["dst"]
        gs, gd = self.G(src), self.G(dst)
        if r["kind"] == "pyramid":        # four-scale head: G[x] += ..., the four 1x1 filters' gradients to their owner
            dw, db = E.pyramid_pool4_bwd(src, r["w"], r["b"], r["k0"], r["slope"], gd, gs)
            r["sink"](grads, dw, db)
        elif r["kind"] == "bn_dropout":   # y = mask * bn(raw): the raw tensor has this one consumer, its gradient is stored
            dg, db = E.bn_dropout_bwd(src, r["mean"], r["var"], r["gamma"], r["eps"], r["mask"], gd, gs)
            bn = r["bn"]
            if bn is not None and bn.weight is not None and bn.weight.requires_grad:
                grad_target(grads, bn.weight).add_(dg)
                grad_target(grads, bn.bias).add_(db)
        else:                             # MaxPool2d(3, 2, 1)(relu(bn(x))): route, then the prologue's backward in place
            pro, bn = r["pro"], r["bn"]
            E.maxpool3s2_bwd(src, pro, gd, gs)
            rows, cpad = E.bn_act_bwd(gs.fd, src.fd, pro, self.ws_bn)
            dg = torch.empty(src.c, dtype=torch.float32, device=src.buf.device)
            dbt = torch.empty(src.c, dtype=torch.float32, device=src.buf.device)
            train_bn = bn.weight is not None and bn.weight.requires_grad
            E.bn_bwd_finalize(self.ws_bn, rows, cpad, src.c, dg, dbt, sink_dgamma=grad_target(grads, bn.weight) if train_bn else None,
                              sink_dbeta=grad_target(grads, bn.bias) if train_bn else None)
            for lo, hi in _constant_entries(pro._meta, src.c):
                dg[lo:hi].zero_()
                dbt[lo:hi].zero_()
            E.bn_bwd_apply(gs.fd, src.fd, pro, dg, dbt, gs.fd, accumulate=False)

    # ---- the whole plan ------------------------------------------------------------------------
    def _hook(self, i):
        if PROGRESS_HOOK is not None:
            self._py(lambda: PROGRESS_HOOK(self, i))      # the hook installed when the step RUNS (a new _Overlap object per step)

    # ---- recording ---------------------------------------------------------------------------------
    def _walk_params(self):
        """Every parameter a walk may add a gradient to, in a fixed order (conv weights, biases, the prologues' BatchNorm pairs)."""
        ps = self.__dict__.get("_walk_param_list")
        if ps is None:
            seen, ps = set(), []
            for i in range(len(self.recs)):
                r = self.recs[i]
                if r["kind"] != "conv":
                    continue
                cand = [r["w"].param, r.get("bias")]
                bn = r["pro"]._meta.get("bn") if r.get("pro") is not None else None
                if bn is not None:
                    cand += [bn.weight, bn.bias]
                for q in cand:
                    if q is not None and id(q) not in seen:
                        seen.add(id(q))
                        ps.append(q)
            self._walk_param_list = ps
        return ps

    def _tape_key(self, skip_dx_of, head):
        ps = self._walk_params() + ([q for q in (head[0]["w"].param, head[0].get("bias")) if q is not None] if head is not None else [])
        return (tuple(q.requires_grad for q in ps), frozenset(skip_dx_of), PROGRESS_HOOK is not None, head is not None,
                self.relu_premask)

    def _sinks_attached(self, params):
        for q in params:
            if q.requires_grad and grad_sink(q) is None:
                return False
        return True

    def _head_first_writer(self, head):
        """The head conv (a network's last one, outside the plan) is the only reader of its input buffer in every network here: it
        then STORES its data gradient and that buffer's gradient (142 MB for the Fusion-discriminator, three walks per step) is
        neither zeroed per walk nor read.  Decided at the first walk that brings the head; a later walk without it zeroes the buffer."""
        hs = self.__dict__.get("_head_store")
        if head is None:
            if hs:
                E.fill_zero(hs[1])
            return
        r = head[0]
        if hs is None:
            self._head_store = hs = ()
            rx = _region(r["x"])
            g = self.gbuf.get(rx[0])
            meta = r["pro"]._meta if r.get("pro") is not None else dict(pool=False)
            others = False
            for rk in self.recs:
                v = rk.get("x") if rk["kind"] == "conv" else rk.get("src")
                if v is not None and hasattr(v, "buf") and _overlap(_region(v), rx):
                    others = True
            ok = (self.fuse_mask and self.defer_affine and os.environ.get("FDGAN_NO_DX_STORE") is None and g is not None and not others
                  and r["stride"] == 1 and not meta["pool"] and r["x"].c0 % 8 == 0 and rx[0] not in self.multi_version
                  and sum(1 for q in self.gbuf.values() if q is g) == 1)
            if ok:
                r["_sole"] = True
                self.nozero.add(id(g))
                self._zero_tables.clear()
                self._head_store = hs = (id(r), g)
        elif hs and hs[0] != id(r):      # another head record on the same plan: back to zero-and-add for good
            self.nozero.discard(id(hs[1]))
            self._zero_tables.clear()
            E.fill_zero(hs[1])
            self._head_store = ()

    def run(self, grads, skip_dx_of=(), head=None):
        """Walks the records in reverse.  The caller has zeroed G and seeded the gradient of the plan's outputs.  `grads`: dict
        parameter -> fp32 gradient, filled / accumulated.  head = (record, dy_view): a conv outside the plan whose backward
        comes first (a network's last conv, launched per call because its output is a fresh tensor); dy_view must live in a
        `persistent` buffer.  From the third walk on (same key) the walk is a recorded tape."""
        self._head_first_writer(head)
        # a recorded walk replays raw pointers and assumes clean accumulation buffers: the caller's zero_() is part of its contract
        # (nothing enforced it; r5: enforced for eager walks too -- the same caller bug must not pass in debug mode and
        # raise in production mode)
        if self._closed:
            raise RuntimeError("PlanBackward.run() after close()")
        if not self._zeroed:
            raise RuntimeError("PlanBackward.run() without a zero_() since the previous walk")
        self._zeroed = False
        key = None
        if self.tape_enabled and not FORCE_EAGER and self.checks is None and self.walks_done >= 2:
            key = self._tape_key(skip_dx_of, head)
            tape = self.tapes.get(key)
            if tape is None and key not in self.tapes:
                params = self._walk_params() + ([q for q in (head[0]["w"].param, head[0].get("bias")) if q is not None] if head is not None else [])
                if self._sinks_attached(params):
                    tape = self._record(grads, skip_dx_of, head)
                self.tapes[key] = tape            # None: this configuration stays eager (a gradient without an optimizer sink)
            if tape is not None:
                for q, ptr in tape.sinks:         # the recorded launches write here: the optimizer must still own these views
                    g = q.grad
                    if g is None or g.data_ptr() != ptr:
                        tape = None
                        break
            if tape is not None:
                return self._replay(tape, grads)
        return self._walk(grads, skip_dx_of, head)

    def _record(self, grads, skip_dx_of, head):
        """A dry walk: every launch lands in the tape's plans instead of on the GPU; host state ends as after a real walk."""
        tape = self._rec = _Tape()
        probe = {}
        ok = False
        try:
            tape._plan()
            self._walk(probe, skip_dx_of, head)
            tape._close()
            ok = all(v is IN_PLACE for v in probe.values())
        finally:
            self._rec = None
            if not ok:
                tape.abort()
                for d in self.deferred.values():
                    d["dirty"].clear()
                self._w_pending = False
        if not ok:
            return None
        self.walks_done -= 1                      # the dry walk computed nothing
        tape.inplace = list(probe.keys())
        tape.sinks = [(q, q._fd_grad_sink.data_ptr()) for q in probe]
        return tape

    def _replay(self, tape, grads):
        cur = torch.cuda.current_stream(self.plan.device).cuda_stream
        streams = (C.c_void_p * 2)(cur, self.wstream.cuda_stream if self.wstream is not None else cur)
        self._w_pending = False                 # raised by the tape after each segment with side-stream launches: a hook that
        tape.replay(streams, self)              # hands a slice to the all-reduce joins the side stream itself (join_side)
        self._w_pending = False                 # the tape ends with the recorded join
        for q in tape.inplace:
            grads[q] = IN_PLACE
        self.walks_done += 1

    def _walk(self, grads, skip_dx_of=(), head=None):
        self.reduce_jobs = []
        self.tr_jobs = []
        if head is not None:
            self.conv_backward(head[0], head[1], grads)
        for i in range(len(self.recs) - 1, -1, -1):
            r = self.recs[i]
            r["_idx"] = i
            if r["kind"] == "copy":
                self.flush(r["dst"])
                E.grad_ew(E.GRAD_ADD, self.G(r["dst"]), self.G(r["src"]))
                continue
            if r["kind"] == "maxpool":
                self.flush(r["dst"])
                E.maxpool2_bwd(r["src"], self.G(r["dst"]), self.G(r["src"]))
                continue
            if r["kind"] == "dropout":      # y = x * mask in place: the gradient takes the same multiply
                self.flush(r["dst"])
                E.mul_mask(r["mask"], self.G(r["dst"]), r["up2"])
                continue
            if r["kind"] in ("pyramid", "bn_dropout", "maxpool3"):      # the legacy DCPDN networks' own ops (csrc/legacy_bwd.hip)
                self.flush(r["dst"])
                self._legacy_backward(r, grads)
                self._hook(i)
                continue
            if i in self.recompute:
                self.recs[self.recompute[i]]["rerun"]()
            y = r["y"]
            need_dx = id(r["x"].buf) not in skip_dx_of and r["x"].buf.data_ptr() not in skip_dx_of
            if y is None:      # a conv that stores a network output (NCHW fp32): the caller seeded its gradient (`_dy`, already
                self.conv_backward(r, r["_dy"], grads, need_dx=need_dx)      # through the output activation's derivative)
                self._hook(i)
                continue
            pending = self._pending_for_fused(r, need_dx)
            gy = None
            if pending is None:
                gy = self._flush_compact(i, r, y)
                if gy is None:
                    self.flush(y)
            if gy is None:
                gy = self.G(y)
            dyv = gy
            if r["upsample"]:
                n, h2, w2, _ = y.shape
                t = self._new_grad(n, h2 // 2, w2 // 2, _r8(y.c))
                dyv = E.View(t, 0, y.c)
                E.grad_ew(E.GRAD_SUMPOOL, gy, dyv)
            r["_post_relu"] = self.x_post_relu.get(i, False)
            if r["e_act"] == L.ACT_RELU:
                if not self.relu_premask:
                    E.grad_ew(E.GRAD_RELU_MASK, dyv, dyv, ref=y)
            elif r["e_act"] == L.ACT_LEAKY02:
                E.grad_ew(E.GRAD_LEAKY_MASK, dyv, dyv, ref=y)
            elif r["e_act"] != L.ACT_NONE:
                raise NotImplementedError("epilogue activation %d inside a plan" % r["e_act"])
            self.conv_backward(r, dyv, grads, need_dx=need_dx, dy_pending=pending)
            if y.buf.data_ptr() in self.multi_version and (self.checks is not None or id(self.gbuf[y.buf.data_ptr()]) not in self.nozero):
                E.fill_zero(self.gbuf[y.buf.data_ptr()])      # the next (earlier) layer accumulates a fresh gradient here
            self._hook(i)
        self.flush_all()      # plan inputs: their gradients are read by the caller
        if self.reduce_jobs:  # the fused bottlenecks' weight gradients: one launch for all of them
            key = tuple((pt.data_ptr(), o.data_ptr(), n, s_, a) for pt, o, n, s_, a in self.reduce_jobs)
            if self.reduce_table is None or self.reduce_table.key != key:
                self.reduce_table = E.ReduceTable(self.reduce_jobs, self.plan.device)
            if self._rec is not None:         # the recorded launch carries the table's device pointer: the tape owns the table
                self._rec.keep.append(self.reduce_table)      # (a later walk under another key may replace it)
            self.reduce_table.launch()
        if self.tr_jobs:      # the side stream's weight gradients: their partial sums, one launch (on that stream, behind them)
            key = tuple((j.part, j.out, j.item_stride, j.items, j.accumulate) for j in self.tr_jobs)
            if self.tr_table is None or self.tr_table.key != key:
                self.tr_table = E.TrReduceTable(self.tr_jobs, list(self.tr_parts.values()), self.plan.device)
            if self._rec is not None:
                self._rec.keep.append(self.tr_table)
            with self._side():
                self.tr_table.launch()
            self._w_pending = True
        if self._deferred_idx is None and self.checks is None:
            self._deferred_idx = frozenset(self._fused_seen) if self.defer_reduce else frozenset()
            if self.defer_reduce:
                self._tr_deferred = frozenset(i for i, n in self._tr_seen.items() if n <= TR_DEFER_MAX)
                for i in self._tr_deferred:
                    self.tr_parts[i] = torch.empty(self._tr_seen[i], dtype=torch.float32, device=self.plan.device)
                    self._share(self.tr_parts[i])
        self.walks_done += 1
        self._hook(len(self.recs))
        self.join_side()      # parameter gradients written on the side stream are complete for whatever follows on this one
