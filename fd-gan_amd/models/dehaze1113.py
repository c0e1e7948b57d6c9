"""MI355X-native `models.dehaze1113`: the FD-GAN generator `FDGAN`, the
Fusion-discriminator `D` and the decoder blocks, behind the reference's nn.Module
surface (class names, constructor signatures, forward I/O, state_dict keys --
/root/reference/models/dehaze1113.py:188-230, :256-275, :358-370, :702-801).

The modules own fp32 master parameters exactly like the reference; `forward` does not
call any torch.nn math.  It builds, per input shape, a static plan of hand-written
gfx950 kernels (libfdgan_hip.so, include/fdgan_hip.h): NHWC bf16 implicit-GEMM
convolutions on MFMA with BatchNorm(+ReLU/LeakyReLU), 2x2 average pooling, concat,
nearest upsampling, bias, tanh/sigmoid folded into their prologues/epilogues.
Inputs must live on the GPU; there is no CPU fallback.
"""
import torch
import torch.nn as nn

from fdgan_hip import engine as E
from fdgan_hip import lib as L
from fdgan_hip.backward import PlanBackward, autograd_grads
from fdgan_hip.netplan import ChanStats, NetPlan, bn_flags

from . import tv_densenet121 as _tv
from fdgan_hip.backward import grad_target


class _PlannedModule(nn.Module):
    """Caches one NetPlan per (shape, BN train/eval flags, parameter storage), least-recently-used first out: a plan owns
    its whole activation set (and, once trained through, a PlanBackward workspace), so the cache is bounded."""
    MAX_PLANS = 8

    def _plan_for(self, x):
        E.require_gpu(x, type(self).__name__ + ".forward")
        if x.dim() != 4:
            raise ValueError("expected a BxCxHxW tensor, got %s" % (tuple(x.shape),))
        cache = self.__dict__.setdefault("_plans", {})
        # a plan built for training keeps every intermediate the backward reads (2.8 GB more at 16 x 256 x 256: nothing
        # next to 288 GB); an inference plan reuses one bottleneck buffer per dense block
        keep = self.__dict__.get("_in_autograd", False) or _wants_grad(self, x)    # grad mode is off inside Function.forward
        key = (tuple(x.shape), x.device.index, bn_flags(self), keep)
        # plans whose parameters were moved / re-allocated (FlatAdam re-points them into its flat buffer, .to()) can never
        # be used again: drop them and their activation sets instead of keeping them behind a dead key
        for k_ in [k_ for k_, pl in cache.items() if pl.param_ptrs() != pl._built_ptrs]:
            cache.pop(k_).close()
        plan = cache.pop(key, None)                       # re-inserted below: the dict's order is the LRU order
        if plan is None:
            for p in self.parameters():
                if p.device != x.device:
                    raise RuntimeError("module parameters are on %s but the input is on %s" % (p.device, x.device))
            self.__dict__["_plan_keep"] = keep
            plan = self._build_plan(tuple(x.shape), x.device)
            plan._built_ptrs = plan.param_ptrs()
            while len(cache) >= self.MAX_PLANS:           # least recently used first; close(): side streams joined, buffers freed HERE
                cache.pop(next(iter(cache))).close()
        cache[key] = plan
        self.__dict__.setdefault("_last_plan", {})[key[:3]] = key     # a key, not the plan: eviction must free it
        plan.forced_dropout = self.__dict__.get("_forced_dropout_masks")     # tests: the masks the oracle drew (NetPlan.refresh_dropout)
        return plan

    def release_plans(self):
        """Closes and drops every cached plan (activation sets, backward workspaces, recorded tapes): deterministic teardown."""
        for pl in self.__dict__.pop("_plans", {}).values():
            pl.close()
        self.__dict__.pop("_last_plan", None)

    def _apply(self, fn, *a, **k):                        # .cuda()/.to(): storages change
        self.release_plans()
        self.__dict__.pop("_last_plan", None)
        self.__dict__.pop("_fd_param_list", None)
        return super()._apply(fn, *a, **k)

    def hip_plan(self, x):
        """The NetPlan serving inputs shaped like `x` (for benchmarks / profiling): the one the last forward of that
        shape ran (training and inference plans differ in what they keep), else a new one for the current grad mode."""
        last = self.__dict__.get("_last_plan", {}).get((tuple(x.shape), x.device.index, bn_flags(self)))
        last = self.__dict__.get("_plans", {}).get(last) if last is not None else None
        if last is not None and last.param_ptrs() == last._built_ptrs:
            return last
        return self._plan_for(x)


# ---------------------------------------------------------------------------------------
# decoder blocks
# ---------------------------------------------------------------------------------------
class _PlanFunction(torch.autograd.Function):
    """autograd bridge of a planned module: forward = the recorded HIP plan, backward = the plan walked in
    reverse (fdgan_hip/backward.py).  Parameters are passed as inputs so autograd routes their gradients."""

    @staticmethod
    def forward(ctx, module, x, *params):
        module.__dict__["_in_autograd"] = True
        try:
            out, state = module._autograd_forward(x)
        finally:
            module.__dict__["_in_autograd"] = False
        ctx.module, ctx.state, ctx.params = module, state, params
        ctx.plan = state[0] if isinstance(state, tuple) else state
        ctx.gen = _bump_generation(ctx.plan)
        return out

    @staticmethod
    def backward(ctx, dout):
        _check_generation(ctx.plan, ctx.gen)
        dx, grads = ctx.module._autograd_backward(ctx.state, dout.detach().float().contiguous())
        return (None, dx) + autograd_grads(grads, ctx.params)


def _bump_generation(plan):
    plan._generation = getattr(plan, "_generation", 0) + 1
    return plan._generation


def _check_generation(plan, gen):
    """The activations live in the plan's buffers, not in autograd: a later forward of the same module and shape
    overwrites them.  Backward must run before the next forward (accumulate losses over several backward calls)."""
    if getattr(plan, "_generation", gen) != gen:
        raise RuntimeError("this module ran another forward (same input shape) before backward: its plan buffers hold "
                           "the later activations. Call backward() after each forward; gradients accumulate in .grad")


def _wants_grad(module, x):
    return torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in _param_list(module)))


def _param_list(module):
    """module.parameters() as a list found once per module (the generator has 786 of them; the recursive walk cost 0.4 ms per
    call).  Validated on every call against the module tree it was built from: every submodule's child and parameter COUNT, and
    the identity of every Parameter object in its owner's `_parameters` dict -- so `m.sub.weight = nn.Parameter(...)`,
    pruning / spectral-norm reparametrisation, or a parameter added to a nested child all rebuild it (the first four
    parameters and the top-level child count alone missed those)."""
    c = module.__dict__.get("_fd_param_list")
    if c is not None:
        for m, n_par, n_mod in c[0]:
            if len(m._parameters) != n_par or len(m._modules) != n_mod:
                c = None
                break
    if c is not None:
        for d, name, p in c[1]:
            if d.get(name) is not p:
                c = None
                break
    if c is None:
        mods = list(module.modules())
        owners = [(m._parameters, name, p) for m in mods for name, p in m._parameters.items() if p is not None]
        c = module.__dict__["_fd_param_list"] = ([(m, len(m._parameters), len(m._modules)) for m in mods], owners,
                                                 list(module.parameters()))
    return c[2]


def _apply_plan_function(module, x):
    params = tuple(p for p in _param_list(module) if p.requires_grad)
    return _PlanFunction.apply(module, x, *params)


def _plan_backward(P):
    if getattr(P, "_closed", False):
        raise RuntimeError("backward through a released plan: the module's plans were closed (release_plans(), .to(), eviction) after "
                           "this forward -- its activation set is gone; run the forward again")
    if getattr(P, "_bwd", None) is None:
        P._bwd = PlanBackward(P)
    return P._bwd


class BottleneckBlockdy(_PlannedModule):
    """dehaze1113.py:256-275.  cat([relu(x), conv2(relu(conv1(relu(x))))], 1); the
    reference's in-place ReLU also overwrites the caller's `x`, reproduced here.
    bn1/bn2 are registered but unused, as in the reference (:260,:264)."""

    def __init__(self, in_planes, out_planes, dropRate=0.0):
        super().__init__()
        inter = out_planes * 4
        self.bn1 = nn.BatchNorm2d(in_planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv1 = nn.Conv2d(in_planes, inter, 1, 1, 0, bias=False)
        self.bn2 = nn.BatchNorm2d(inter)
        self.conv2 = nn.Conv2d(inter, out_planes, 3, 1, 1, bias=False)
        self.droprate = dropRate          # > 0: F.dropout after each conv in training mode (:270-274; FDGAN passes the default 0)
        self.in_planes, self.inter, self.out_planes = in_planes, inter, out_planes

    def emit(self, P, blk, tmp):
        """blk: View of the (in+out)-channel buffer whose first `in_planes` channels hold x."""
        cin, cout = self.in_planes, self.out_planes
        w1 = P.weight(self.conv1.weight, self.inter, cin, 1)
        w2 = P.weight(self.conv2.weight, cout, self.inter, 3)
        relu = E.make_prologue(act=L.ACT_RELU)
        drop = self.droprate > 0 and self.training
        P.conv(E.View(blk.buf, blk.c0, cin), w1, tmp, 1, pro=relu)
        if drop:
            P.dropout(tmp, self.droprate)
        P.conv(tmp, w2, E.View(blk.buf, blk.c0 + cin, cout), 3, pad=1, pro=relu)
        if drop:
            P.dropout(E.View(blk.buf, blk.c0 + cin, cout), self.droprate)

    def _build_plan(self, shape, dev):
        n, c, h, w = shape
        if c != self.in_planes:
            raise ValueError("expected %d input channels, got %d" % (self.in_planes, c))
        P = NetPlan(dev)
        P.blk = E.new_act(n, h, w, c + self.out_planes, dev)
        P.tmp = E.new_act(n, h, w, self.inter, dev)
        self.emit(P, E.View(P.blk), E.View(P.tmp))
        return P.finish()

    def forward(self, x):
        if _wants_grad(self, x):
            if not x.requires_grad:
                x.relu_()                                           # reference aliasing (:261) for plain data
            return _apply_plan_function(self, x)
        P = self._plan_for(x)
        with torch.no_grad():
            x.relu_()                                               # reference aliasing (:261)
            E.to_nhwc(x.detach().float().contiguous(), E.View(P.blk, 0, self.in_planes))
            P.launch()
            out = torch.empty((x.shape[0], self.in_planes + self.out_planes) + tuple(x.shape[2:]),
                              dtype=torch.float32, device=x.device)
            E.to_nchw(E.View(P.blk), out)
        return out

    # under autograd a tensor that itself requires grad is NOT overwritten (same values and gradients as the
    # reference's in-place ReLU; only the aliasing side effect is dropped -- PyTorch would refuse it on a leaf)
    def _autograd_forward(self, x):
        P = self._plan_for(x)
        E.to_nhwc(torch.relu(x.detach().float()).contiguous(), E.View(P.blk, 0, self.in_planes))
        P.launch()
        out = torch.empty((x.shape[0], self.in_planes + self.out_planes) + tuple(x.shape[2:]), dtype=torch.float32,
                          device=x.device)
        E.to_nchw(E.View(P.blk), out)
        return out, P

    def _autograd_backward(self, P, dout):
        B = _plan_backward(P)
        B.zero_()
        blk = E.View(P.blk)
        E.to_nhwc(dout, B.G(blk))
        grads = {}
        B.run(grads)
        gx = B.G(E.View(P.blk, 0, self.in_planes))
        E.grad_ew(E.GRAD_RELU_MASK, gx, gx, ref=E.View(P.blk, 0, self.in_planes))      # d relu(x) / dx
        dx = torch.empty((dout.shape[0], self.in_planes) + tuple(dout.shape[2:]), dtype=torch.float32, device=dout.device)
        E.to_nchw(gx, dx)
        return dx, grads


class TransitionBlockdy(_PlannedModule):
    """dehaze1113.py:358-370: relu (in place) -> ConvTranspose2d 1x1 (weight
    (Cin,Cout,1,1), no bias) -> nearest x2 upsample."""

    def __init__(self, in_planes, out_planes, dropRate=0.0):
        super().__init__()
        self.bn1 = nn.BatchNorm2d(in_planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv1 = nn.ConvTranspose2d(in_planes, out_planes, 1, 1, 0, bias=False)
        self.droprate = dropRate          # > 0: F.dropout between the conv and the upsample in training mode (:367-368)
        self.in_planes, self.out_planes = in_planes, out_planes

    def emit(self, P, x, y):
        w = P.weight(self.conv1.weight, self.out_planes, self.in_planes, 1, transposed=True)
        P.conv(x, w, y, 1, pro=E.make_prologue(act=L.ACT_RELU), upsample=True)
        if self.droprate > 0 and self.training:
            P.dropout(y, self.droprate, up2=True)       # the mask is drawn at the conv's resolution, before the nearest x2

    def _build_plan(self, shape, dev):
        n, c, h, w = shape
        if c != self.in_planes:
            raise ValueError("expected %d input channels, got %d" % (self.in_planes, c))
        P = NetPlan(dev)
        P.xin = E.new_act(n, h, w, (c + 7) // 8 * 8, dev)
        P.yout = E.new_act(n, 2 * h, 2 * w, (self.out_planes + 7) // 8 * 8, dev)
        self.emit(P, E.View(P.xin, 0, c), E.View(P.yout, 0, self.out_planes))
        return P.finish()

    def forward(self, x):
        if _wants_grad(self, x):
            if not x.requires_grad:
                x.relu_()
            return _apply_plan_function(self, x)
        P = self._plan_for(x)
        with torch.no_grad():
            x.relu_()
            E.to_nhwc(x.detach().float().contiguous(), E.View(P.xin))
            P.launch()
            out = torch.empty((x.shape[0], self.out_planes, 2 * x.shape[2], 2 * x.shape[3]), dtype=torch.float32,
                              device=x.device)
            E.to_nchw(E.View(P.yout, 0, self.out_planes), out)
        return out

    def _autograd_forward(self, x):
        P = self._plan_for(x)
        E.to_nhwc(torch.relu(x.detach().float()).contiguous(), E.View(P.xin))
        P.launch()
        out = torch.empty((x.shape[0], self.out_planes, 2 * x.shape[2], 2 * x.shape[3]), dtype=torch.float32, device=x.device)
        E.to_nchw(E.View(P.yout, 0, self.out_planes), out)
        return out, P

    def _autograd_backward(self, P, dout):
        B = _plan_backward(P)
        B.zero_()
        E.to_nhwc(dout, B.G(E.View(P.yout)))
        grads = {}
        B.run(grads)
        xin = E.View(P.xin, 0, self.in_planes)
        gx = B.G(xin)
        E.grad_ew(E.GRAD_RELU_MASK, gx, gx, ref=xin)
        dx = torch.empty((dout.shape[0], self.in_planes, dout.shape[2] // 2, dout.shape[3] // 2), dtype=torch.float32,
                         device=dout.device)
        E.to_nchw(gx, dx)
        return dx, grads


# ---------------------------------------------------------------------------------------
# generator
# ---------------------------------------------------------------------------------------
def _emit_dense_block(P, block, blk_buf, stats, bott_buf, count, keep=False):
    """torchvision _DenseBlock on a pre-allocated concat buffer: layer i reads channels
    [0, cin_i) and writes its 32 new channels at [cin_i, cin_i+32).  keep: every layer gets its own bottleneck
    buffer and statistics (training: the backward reads them instead of re-running the 1x1 conv); their gradients
    still share one buffer, each being consumed before the next is produced."""
    bott = E.View(bott_buf)
    bstats = ChanStats(_tv.BN_SIZE * _tv.GROWTH, P.device)
    P.keep.append(bstats)
    cin = block.cin
    for li, layer in enumerate(block.values()):
        if keep and li > 0:
            own = torch.empty_like(bott_buf)
            P.keep.append(own)
            P.grad_alias[own.data_ptr()] = bott_buf.data_ptr()
            bott = E.View(own)
            bstats = ChanStats(_tv.BN_SIZE * _tv.GROWTH, P.device)
            P.keep.append(bstats)
        w1 = P.weight(layer.conv1.weight, _tv.BN_SIZE * _tv.GROWTH, cin, 1)
        w2 = P.weight(layer.conv2.weight, _tv.GROWTH, _tv.BN_SIZE * _tv.GROWTH, 3)
        P.conv(E.View(blk_buf, 0, cin), w1, bott, 1, pro=P.bn_prologue(layer.norm1, stats, count),
               stats=bstats if layer.norm2.training else None)
        P.conv(bott, w2, E.View(blk_buf, cin, _tv.GROWTH), 3, pad=1, pro=P.bn_prologue(layer.norm2, bstats, count),
               stats=stats, stats_c0=cin)
        cin += _tv.GROWTH


def _emit_transition(P, trans, x, stats, y, count, out_stats=None, out_c0=0):
    """torchvision _Transition: BN -> ReLU -> 1x1 conv -> AvgPool2.  The bias-free 1x1
    conv commutes with the average pool, so the pool runs in the prologue (4x fewer MACs)."""
    cin, cout = trans.conv.in_channels, trans.conv.out_channels
    w = P.weight(trans.conv.weight, cout, cin, 1)
    P.conv(x, w, y, 1, pro=P.bn_prologue(trans.norm, stats, count, pool=True), stats=out_stats, stats_c0=out_c0)


class FDGAN(_PlannedModule):
    """dehaze1113.py:702-801.  forward: (B,3,H,W) float in [0,1] -> (B,3,H,W) in (-1,1);
    H and W multiples of 8.  Default mode is train (batch-statistics BatchNorm), which
    the reference also uses for inference (README.md:38)."""

    def __init__(self):
        super().__init__()
        feats = _tv.densenet121(pretrained=True).features
        self.conv0 = feats.conv0                       # registered, never called (:709)
        self.relu0 = feats.relu0
        self.dense_block1 = feats.denseblock1
        self.trans_block1 = feats.transition1
        self.dense_block2 = feats.denseblock2
        self.trans_block2 = feats.transition2
        self.dense_block3 = feats.denseblock3
        self.trans_block3 = feats.transition3
        self.dense_block31 = feats.denseblock4        # registered, never called (:725)
        self.dense_norm31 = feats.norm5                # registered, never called (:728)
        self.dense_block4 = BottleneckBlockdy(512, 256)
        self.trans_block4 = TransitionBlockdy(768, 128)
        self.dense_block5 = BottleneckBlockdy(384, 128)
        self.trans_block5 = TransitionBlockdy(512, 64)
        self.dense_block6 = BottleneckBlockdy(64, 32)
        self.trans_block6 = TransitionBlockdy(96, 16)
        self.conv_refin1 = nn.Conv2d(3, 64, 3, 1, 1)
        self.conv_refin6 = nn.Conv2d(640, 512, 3, 1, 1)
        self.conv_refin5 = nn.Conv2d(256, 128, 1, 1, 0)
        self.tanh = nn.Tanh()
        self.conv_refin3 = nn.Conv2d(16, 3, kernel_size=3, stride=1, padding=1)
        self.conv_refin2 = nn.Conv2d(64, 32, kernel_size=1, stride=1, padding=0)
        self.conv_refine4 = nn.Conv2d(160, 128, kernel_size=3, stride=1, padding=1)

    # The plan mirrors forward() of the reference line by line (:758-801).
    def _build_plan(self, shape, dev):
        n, c, h, w = shape
        if c != 3:
            raise ValueError("FDGAN expects 3 input channels, got %d" % c)
        if h % 8 or w % 8:
            raise ValueError("FDGAN needs H and W to be multiples of 8 (skip concats), got %dx%d" % (h, w))
        P = NetPlan(dev)
        keep = self.__dict__.get("_plan_keep", False)
        h2, w2, h4, w4, h8, w8 = h // 2, w // 2, h // 4, w // 4, h // 8, w // 8
        A = lambda hh, ww, cc: E.new_act(n, hh, ww, cc, dev)
        P.in8 = E.new_act(n, h, w, 8, dev, zero=True)
        # (pixel pitches 512 / 1024 / 2048 B: powers of two.  affine_accumulate alone streams a 256-channel range 55 % faster at a pitch of
        # 2112 B than at 2048 (tools/pitch_probe.py 128 256), so padded concat buffers were tried, round 4: +32 channels 27.9 ms per
        # step against 26.9, +64 channels 27.2 -- the conv kernels' rows want whole aligned 128-byte lines more than the odd pitch helps)
        blk1, bott1, cat1 = A(h, w, 256), A(h, w, 128), A(h2, w2, 160)
        blk2, bott2 = A(h2, w2, 512), A(h2, w2, 128)
        blk3, bott3 = A(h4, w4, 1024), A(h4, w4, 128)
        cat6, blk4, tmp4 = A(h8, w8, 640), A(h8, w8, 768), A(h8, w8, 1024)
        blk5, tmp5 = A(h4, w4, 512), A(h4, w4, 512)
        blk6, tmp6 = A(h2, w2, 96), A(h2, w2, 128)
        x6 = A(h, w, 16)
        P.x6 = x6
        st1, st2, st3 = ChanStats(256, dev), ChanStats(512, dev), ChanStats(1024, dev)
        P.keep += [st1, st2, st3]
        P.taps = {"x0": E.View(blk1, 0, 64), "x01": E.View(cat1, 0, 32), "x1": E.View(cat1, 32, 128),
                  "x10": E.View(blk2, 0, 128), "x2": E.View(blk3, 0, 256), "x3": E.View(cat6, 0, 512),
                  "x22": E.View(cat6, 512, 128), "x4": E.View(blk5, 0, 128), "x5": E.View(blk6, 0, 64),
                  "x6": E.View(x6)}
        cnt1, cnt2, cnt3 = n * h * w, n * h2 * w2, n * h4 * w4
        train = self.training

        # x0 = relu0(conv_refin1(x))                                              (:760)
        P.conv(E.View(P.in8, 0, 3), P.weight(self.conv_refin1.weight, 64, 3, 3), E.View(blk1, 0, 64), 3, pad=1,
               bias=self.conv_refin1.bias, e_act=L.ACT_RELU, stats=st1)
        # x01 = conv_refin2(avg_pool2d(x0, 2))                                    (:763)
        P.conv(E.View(blk1, 0, 64), P.weight(self.conv_refin2.weight, 32, 64, 1), E.View(cat1, 0, 32), 1,
               bias=self.conv_refin2.bias, pro=E.make_prologue(pool=True))
        # x1 = trans_block1(dense_block1(x0))                                     (:767-769)
        _emit_dense_block(P, self.dense_block1, blk1, st1, bott1, cnt1, keep)
        _emit_transition(P, self.trans_block1, E.View(blk1, 0, 256), st1, E.View(cat1, 32, 128), cnt1)
        # x10 = conv_refine4(cat[x01, x1])                                        (:773)
        P.conv(E.View(cat1), P.weight(self.conv_refine4.weight, 128, 160, 3), E.View(blk2, 0, 128), 3, pad=1,
               bias=self.conv_refine4.bias, stats=st2)
        # x2 = trans_block2(dense_block2(x10))                                    (:774)
        _emit_dense_block(P, self.dense_block2, blk2, st2, bott2, cnt2, keep)
        _emit_transition(P, self.trans_block2, E.View(blk2, 0, 512), st2, E.View(blk3, 0, 256), cnt2, out_stats=st3)
        P.copy(E.View(blk3, 0, 256), E.View(blk5, 128, 256))                      # x2 half of x42 (:786)
        # x22 = conv_refin5(avg_pool2d(x2, 2))                                    (:780) -- emitted ahead of dense block 3 (it only needs
        # x2): trans_block3 is then the LAST reader of blk3, i.e. the first writer of its gradient buffer in the backward walk, reads
        # all of it, and stores instead of adding (fdgan_hip/backward.py: first writers) -- no zeroing of the block's gradient buffer
        P.conv(E.View(blk3, 0, 256), P.weight(self.conv_refin5.weight, 128, 256, 1), E.View(cat6, 512, 128), 1,
               bias=self.conv_refin5.bias, pro=E.make_prologue(pool=True))
        # x3 = trans_block3(dense_block3(x2))                                     (:778)
        _emit_dense_block(P, self.dense_block3, blk3, st3, bott3, cnt3, keep)
        _emit_transition(P, self.trans_block3, E.View(blk3, 0, 1024), st3, E.View(cat6, 0, 512), cnt3)
        # x4 = trans_block4(dense_block4(conv_refin6(cat[x3, x22])))              (:783)
        P.conv(E.View(cat6), P.weight(self.conv_refin6.weight, 512, 640, 3), E.View(blk4, 0, 512), 3, pad=1,
               bias=self.conv_refin6.bias)
        self.dense_block4.emit(P, E.View(blk4), E.View(tmp4))
        self.trans_block4.emit(P, E.View(blk4), E.View(blk5, 0, 128))
        # x5 = trans_block5(dense_block5(cat[x4, x2]))                            (:786-790)
        self.dense_block5.emit(P, E.View(blk5), E.View(tmp5))
        self.trans_block5.emit(P, E.View(blk5), E.View(blk6, 0, 64))
        # x6 = trans_block6(dense_block6(x5))                                     (:795)
        self.dense_block6.emit(P, E.View(blk6), E.View(tmp6))
        self.trans_block6.emit(P, E.View(blk6), E.View(x6))
        # dehaze = tanh(conv_refin3(x6)) is launched per call into a fresh output  (:799)
        P.w_last = P.weight(self.conv_refin3.weight, 3, 16, 3)
        P.last_desc = E.conv_desc(3, 1, 1, L.ACT_TANH, False, cout=3)
        del train
        return P.finish()

    def forward(self, x):
        if _wants_grad(self, x):
            return _apply_plan_function(self, x)
        return self._forward_plan(self._plan_for(x), x)

    def _autograd_forward(self, x):
        P = self._plan_for(x)
        out = self._forward_plan(P, x)
        return out, (P, out, bool(x.requires_grad))

    def _autograd_backward(self, state, dout):
        """Gradients of every parameter that receives one (11.8 M of 13.98 M; conv0, dense_block31, dense_norm31
        and the dy blocks' bn1/bn2 never do).  The generator's input is data and its gradient is normally not
        formed (the walk skips conv_refin1's data gradient); an input that requires grad gets it (round 6: until then the
        request was silently answered with None) -- the image's one reader is conv_refin1, whose data gradient the walk then
        leaves in the input buffer's gradient."""
        P, out, need_dx = state      # train- or eval-mode BatchNorm (eval: running statistics are constants, backward.py:_constant_entries)
        B = _plan_backward(P)
        B.zero_()
        n, _, h, w = out.shape
        g8 = B.persistent("g8", lambda: E.new_grad(n, h, w, 8, out.device))      # read by the (recorded) walk's first launches
        E.out_act_bwd(dout, out, L.ACT_TANH, E.View(g8))                          # dehaze = tanh(conv_refin3(x6)) (:799)
        grads = {}
        last = B.persistent("last", lambda: dict(x=E.View(P.x6), w=P.w_last, k=3, pad=1, stride=1, bias=self.conv_refin3.bias, pro=None))
        B.run(grads, skip_dx_of=() if need_dx else {P.in8.data_ptr()}, head=(last, E.View(g8, 0, 3)))
        dx = None
        if need_dx:
            dx = torch.empty((n, 3, h, w), dtype=torch.float32, device=out.device)
            E.to_nchw(B.G(E.View(P.in8, 0, 3)), dx)
        return dx, grads

    def _forward_plan(self, P, x):
        with torch.no_grad():
            E.to_nhwc(x.detach().float().contiguous(), E.View(P.in8))
            P.launch()
            out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
            E.conv2d(E.View(P.x6).fd, P.w_last, self.conv_refin3.bias, None, E.nchw_f32_view(out), P.last_desc)
        return out


# ---------------------------------------------------------------------------------------
# legacy DCPDN network `Dense`: forward, and backward in train mode
# ---------------------------------------------------------------------------------------
class BottleneckBlock(nn.Module):
    """dehaze1113.py:234-254 / dehaze22.py:491-510: BN-ReLU-1x1 (4 x out), BN-ReLU-3x3, concat.  Parameter container with
    `emit`; used inside `Dense`'s plan."""

    def __init__(self, in_planes, out_planes, dropRate=0.0):
        super().__init__()
        if dropRate > 0:
            raise NotImplementedError("dropRate > 0 is never used by the reference's networks")
        inter = out_planes * 4
        self.bn1 = nn.BatchNorm2d(in_planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv1 = nn.Conv2d(in_planes, inter, 1, 1, 0, bias=False)
        self.bn2 = nn.BatchNorm2d(inter)
        self.conv2 = nn.Conv2d(inter, out_planes, 3, 1, 1, bias=False)
        self.droprate = dropRate
        self.in_planes, self.inter, self.out_planes = in_planes, inter, out_planes

    def forward(self, x):
        raise RuntimeError("container only; the math runs in the enclosing network's HIP plan")

    def emit(self, P, buf, stats, tmp, count):
        """buf: (in + out)-channel buffer whose first `in_planes` channels hold x (their statistics in stats[0:in])."""
        cin, cout = self.in_planes, self.out_planes
        tstats = ChanStats(self.inter, P.device)
        P.keep.append(tstats)
        P.conv(E.View(buf, 0, cin), P.weight(self.conv1.weight, self.inter, cin, 1), E.View(tmp), 1,
               pro=P.bn_prologue(self.bn1, stats, count), stats=tstats if self.bn2.training else None)
        P.conv(E.View(tmp), P.weight(self.conv2.weight, cout, self.inter, 3), E.View(buf, cin, cout), 3, pad=1,
               pro=P.bn_prologue(self.bn2, tstats, count), stats=stats, stats_c0=cin)


class TransitionBlock(nn.Module):
    """dehaze1113.py:343-356 / dehaze22.py:512-529: BN-ReLU-ConvTranspose2d 1x1 (weight (Cin, Cout, 1, 1)), nearest x2."""

    def __init__(self, in_planes, out_planes, dropRate=0.0):
        super().__init__()
        if dropRate > 0:
            raise NotImplementedError("dropRate > 0 is never used by the reference's networks")
        self.bn1 = nn.BatchNorm2d(in_planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv1 = nn.ConvTranspose2d(in_planes, out_planes, 1, 1, 0, bias=False)
        self.droprate = dropRate
        self.in_planes, self.out_planes = in_planes, out_planes

    def forward(self, x):
        raise RuntimeError("container only; the math runs in the enclosing network's HIP plan")

    def emit(self, P, buf, stats, count, y, out_stats=None, out_c0=0):
        w = P.weight(self.conv1.weight, self.out_planes, self.in_planes, 1, transposed=True)
        P.conv(E.View(buf), w, y, 1, pro=P.bn_prologue(self.bn1, stats, count), upsample=True, stats=out_stats, stats_c0=out_c0)


def _stem_filter(w7, out):
    """conv0 (64, 3, 7, 7), stride 2, pad 3  ->  (64, 16, 4, 4) stride 1 on the 2x2 space-to-depth image (channel c 4 + dy 2 + dx,
    as F.pixel_unshuffle orders them; 12 used + 4 zero): the 7x7 window is extended by a zero tap in front to 8x8 = 4x4 blocks."""
    out.zero_()
    w8 = torch.zeros((w7.shape[0], w7.shape[1], 8, 8), dtype=w7.dtype, device=w7.device)
    w8[:, :, 1:, 1:] = w7.detach()
    for dy in range(2):
        for dx in range(2):
            out[:, dy * 2 + dx:12:4] = w8[:, :, dy::2, dx::2]
    return out


class _DenseBase(_PlannedModule):
    """The DCPDN `Dense` network (dehaze1113.py:431-570, :572-699; dehaze22.py:531-660), forward and train-mode backward, on the generator's own
    machinery: torchvision's dense blocks and transitions on concat buffers, BatchNorm in the consumers' prologues.  New
    here: the DenseNet stem -- the 7x7 stride-2 conv runs as a 4x4 stride-1 conv on the space-to-depth image (exact: a
    zero-extended 8x8 window), norm0 / relu0 / MaxPool2d(3, 2, 1) as `fdgan_maxpool3s2_nhwc` -- and decoder blocks WITH
    BatchNorm.  H and W must be multiples of 32."""
    TAIL = None

    def _make_layers(self):
        feats = _tv.densenet121(pretrained=True).features
        self.conv0, self.norm0, self.relu0, self.pool0 = feats.conv0, feats.norm0, feats.relu0, feats.pool0
        self.dense_block1, self.trans_block1 = feats.denseblock1, feats.transition1
        self.dense_block2, self.trans_block2 = feats.denseblock2, feats.transition2
        self.dense_block3, self.trans_block3 = feats.denseblock3, feats.transition3
        self.dense_block4, self.trans_block4 = BottleneckBlock(512, 256), TransitionBlock(768, 128)
        self.dense_block5, self.trans_block5 = BottleneckBlock(384, 256), TransitionBlock(640, 128)
        self.dense_block6, self.trans_block6 = BottleneckBlock(256, 128), TransitionBlock(384, 64)
        self.dense_block7, self.trans_block7 = BottleneckBlock(64, 64), TransitionBlock(128, 32)
        self.dense_block8, self.trans_block8 = BottleneckBlock(32, 32), TransitionBlock(64, 16)
        self.conv_refin = nn.Conv2d(19, 20, 3, 1, 1)
        self.tanh = nn.Tanh()
        for nm in ("conv1010", "conv1020", "conv1030", "conv1040"):
            setattr(self, nm, nn.Conv2d(20, 1, kernel_size=1, stride=1, padding=0))

    def _build_plan(self, shape, dev):
        n, c, h, w = shape
        if c != 3:
            raise ValueError("%s expects 3 input channels, got %d" % (type(self).__name__, c))
        if h % 32 or w % 32:
            raise ValueError("%s needs H and W to be multiples of 32, got %dx%d" % (type(self).__name__, h, w))
        train = self.training
        P = NetPlan(dev)
        A = lambda hh, ww, cc, zero=False: E.new_act(n, hh, ww, cc, dev, zero=zero)
        h2, w2, h4, w4, h8, w8, h16, w16, h32, w32 = h // 2, w // 2, h // 4, w // 4, h // 8, w // 8, h // 16, w // 16, h // 32, w // 32
        # ---- stem: conv0 on the space-to-depth image (two zero block rows / columns in front, one behind: pad 3 of the 7x7)
        P.xs = A(h2 + 3, w2 + 3, 16, zero=True)
        P.xs_in = E.StridedView(P.xs, 0, 16, 2, 2, 1, 1, h2, w2)
        P.w0 = torch.zeros((64, 16, 4, 4), dtype=torch.float32, device=dev)
        P.w0.requires_grad_(self.conv0.weight.requires_grad)      # a derived filter: its gradient is mapped back to conv0's (`_derived_grads`)
        c0 = A(h2, w2, 64)
        st0 = ChanStats(64, dev)
        P.conv(E.View(P.xs), P.weight(P.w0, 64, 16, 4), E.View(c0), 4, pad=0, stats=st0 if train else None, label="conv0")
        blk1, bott1 = A(h4, w4, 256), A(h4, w4, 128)
        st1 = ChanStats(256, dev)
        pro0 = P.bn_prologue(self.norm0, st0, n * h2 * w2, act=L.ACT_RELU)
        pro0 = E.prologue_without_side_effects(pro0)          # norm0's running statistics: updated after the launch (`_run`)
        pool_info = L.FdConvInfo()
        pool_rows = (n * h4 * w4 + 31) // 32
        pool_info.stats_rows, pool_info.stats_cpad = pool_rows, 64
        x0 = E.View(blk1, 0, 64)
        P._ops.append((lambda: (E.maxpool3s2(E.View(c0), pro0, x0, P.ws if train else None),
                                E.bn_finalize(P.ws, pool_info, 64, n * h4 * w4, st1.mean, st1.var, 0) if train else None),
                       pool_rows * 64 * 2 if train else 0, dict(label="pool0", flops=0.0, flops_done=0.0, bytes=n * h2 * w2 * 64 * 2 * 5 // 4)))
        P.records.append(dict(kind="maxpool3", src=E.View(c0), dst=x0, pro=pro0, bn=self.norm0))
        P.st0, P.cnt0 = st0, n * h2 * w2
        # ---- encoder
        blk2, bott2 = A(h8, w8, 512), A(h8, w8, 128)
        blk3, bott3 = A(h16, w16, 1024), A(h16, w16, 128)
        st2, st3 = ChanStats(512, dev), ChanStats(1024, dev)
        b4, t4 = A(h32, w32, 768), A(h32, w32, 1024)
        b5, t5 = A(h16, w16, 640), A(h16, w16, 1024)
        b6, t6 = A(h8, w8, 384), A(h8, w8, 512)
        b7, t7 = A(h4, w4, 128), A(h4, w4, 256)
        b8, t8 = A(h2, w2, 64), A(h2, w2, 128)
        s4, s5, s6, s7, s8 = (ChanStats(cc, dev) for cc in (768, 640, 384, 128, 64))
        cnt4, cnt8, cnt16, cnt32, cnt2 = n * h4 * w4, n * h8 * w8, n * h16 * w16, n * h32 * w32, n * h2 * w2
        _emit_dense_block(P, self.dense_block1, blk1, st1, bott1, cnt4)
        # x1 feeds dense_block2 AND the concat x52 = [x5, x1] (:617): stored once, copied, statistics finalized into both tables
        tb = self.trans_block1
        P.conv(E.View(blk1), P.weight(tb.conv.weight, 128, 256, 1), E.View(blk2, 0, 128), 1, pro=P.bn_prologue(tb.norm, st1, cnt4, pool=True),
               stats=st2 if train else None, stats_also=((s6, 128),) if train else ())
        P.copy(E.View(blk2, 0, 128), E.View(b6, 128, 128))
        _emit_dense_block(P, self.dense_block2, blk2, st2, bott2, cnt8)
        tb = self.trans_block2      # x2: dense_block3's input and the concat x42 = [x4, x2] (:612)
        P.conv(E.View(blk2), P.weight(tb.conv.weight, 256, 512, 1), E.View(blk3, 0, 256), 1, pro=P.bn_prologue(tb.norm, st2, cnt8, pool=True),
               stats=st3 if train else None, stats_also=((s5, 128),) if train else ())
        P.copy(E.View(blk3, 0, 256), E.View(b5, 128, 256))
        _emit_dense_block(P, self.dense_block3, blk3, st3, bott3, cnt16)
        _emit_transition(P, self.trans_block3, E.View(blk3, 0, 1024), st3, E.View(b4, 0, 512), cnt16, out_stats=s4 if train else None)
        # ---- decoder
        self.dense_block4.emit(P, b4, s4, t4, cnt32)
        self.trans_block4.emit(P, b4, s4, cnt32, E.View(b5, 0, 128), out_stats=s5 if train else None)
        self.dense_block5.emit(P, b5, s5, t5, cnt16)
        self.trans_block5.emit(P, b5, s5, cnt16, E.View(b6, 0, 128), out_stats=s6 if train else None)
        self.dense_block6.emit(P, b6, s6, t6, cnt8)
        self.trans_block6.emit(P, b6, s6, cnt8, E.View(b7, 0, 64), out_stats=s7 if train else None)
        self.dense_block7.emit(P, b7, s7, t7, cnt4)
        self.trans_block7.emit(P, b7, s7, cnt4, E.View(b8, 0, 32), out_stats=s8 if train else None)
        self.dense_block8.emit(P, b8, s8, t8, cnt2)
        P.cat8 = A(h, w, 24, zero=True)                                   # [x8 (16) | x (3) | 5 zero]
        self.trans_block8.emit(P, b8, s8, cnt2, E.View(P.cat8, 0, 16))
        P.out = torch.empty((n, 3, h, w), dtype=torch.float32, device=dev)
        P.copies = []
        self._build_tail(P, n, h, w, dev, train)
        P.keep += [c0, st0, st1, st2, st3, s4, s5, s6, s7, s8, blk1, blk2, blk3, b4, b5, b6, b7, b8, t4, t5, t6, t7, t8, bott1, bott2, bott3, x0, pro0]
        return P.finish()

    def forward(self, x):
        if _wants_grad(self, x):
            return _apply_plan_function(self, x)
        return self._forward_plan(x)[1]

    def _autograd_forward(self, x):
        P, out = self._forward_plan(x)
        return out, (P, out, bool(x.requires_grad))

    def _autograd_backward(self, state, dout):
        """torch.autograd through dehaze1113.py:431-570 / :572-699 (dehaze22.py:531-660), train- or eval-mode BatchNorm: the recorded plan
        walked in reverse (fdgan_hip/backward.py) -- the encoder's and decoder's dense blocks on the generator's own backward
        kernels, the stem's MaxPool2d(3, 2, 1) and the four-scale head on csrc/legacy_bwd.hip."""
        P, out, need_dx = state
        B = _plan_backward(P)
        B.zero_()
        n, _, h, w = out.shape
        g8 = E.new_grad(n, h, w, 8, out.device)
        E.out_act_bwd(dout, out, L.ACT_TANH, E.View(g8))                   # tanh(refine3(.)) (:569, :698)
        last = P.records[-1]
        assert last["kind"] == "conv" and last["y"] is None
        last["_dy"] = E.View(g8, 0, 3)
        grads = {}
        B.run(grads, skip_dx_of={P.xs.data_ptr()})
        self._derived_grads(P, grads)
        dx = None
        if need_dx:
            # round 6: the input image has two readers -- conv0 (7x7 stride 2, pad 3: here on the space-to-depth copy, so its data
            # gradient is taken with the parameter's own 7x7 filter on the any-stride direct kernel) and the concatenation in front
            # of conv_refin (:552 / :681), whose gradient the walk left in channels 16..18 of cat8's buffer
            rec0 = next(r for r in P.records if r["kind"] == "conv" and r["x"].buf is P.xs)
            dx = torch.empty((n, 3, h, w), dtype=torch.float32, device=out.device)
            E.conv_bwd_data_direct(B.G(rec0["y"]).fd, self.conv0.weight.detach().contiguous(), E.conv_desc(7, 2, 3, cout=64), dx)
            dx += B.G(E.View(P.cat8, 16, 3)).torch_nchw()
        return dx, grads

    def _derived_grads(self, P, grads):
        """Gradients of the filters the plan derives from parameters, mapped back to those parameters."""
        g0 = grads.pop(P.w0, None)
        if g0 is not None:                                                  # inverse of _stem_filter
            g8 = torch.zeros((64, 3, 8, 8), dtype=torch.float32, device=g0.device)
            for dy in range(2):
                for dx in range(2):
                    g8[:, :, dy::2, dx::2] = g0[:, dy * 2 + dx:12:4]
            grads[self.conv0.weight] = g8[:, :, 1:, 1:].contiguous()

    def _forward_plan(self, x):
        P = self._plan_for(x)
        w7 = self.conv0.weight
        P.w0.requires_grad_(w7.requires_grad)             # a derived leaf follows its source: freezing / unfreezing after the plan exists
        with torch.no_grad():
            if getattr(P.w0, "_src_version", None) != (w7._version, w7.data_ptr()):
                _stem_filter(w7, P.w0)
                P.w0._src_version = (w7._version, w7.data_ptr())
            for dst, src_t in P.copies:
                dst.copy_(src_t.detach())
            self._refresh_tail(P)
            xf = x.detach().float().contiguous()
            E.to_nhwc(torch.nn.functional.pixel_unshuffle(xf, 2).contiguous(), P.xs_in)
            E.to_nhwc(xf, E.View(P.cat8, 16, 8))
            P.launch()
            if self.training:       # norm0 is consumed by the pooling kernel, which has no side effects
                bn, m = self.norm0, self.norm0.momentum if self.norm0.momentum is not None else 0.1
                bn.running_mean.mul_(1 - m).add_(P.st0.mean, alpha=m)
                bn.running_var.mul_(1 - m).add_(P.st0.var, alpha=m * P.cnt0 / max(P.cnt0 - 1, 1))
                bn.num_batches_tracked.add_(1)
            return P, P.out.clone()


def _pyramid_sink(convs):
    """Gradient sink of a four-scale head's 1x1 filters (backward.py, record kind "pyramid")."""
    def sink(grads, dw, db):
        for i, conv in enumerate(convs):
            if conv.weight.requires_grad:
                grad_target(grads, conv.weight).add_(dw[i].view_as(conv.weight))
            if conv.bias is not None and conv.bias.requires_grad:
                grad_target(grads, conv.bias).add_(db[i:i + 1])
    return sink


def _permuted_final_grad(P, grads, conv):
    """The final filter of a four-scale head reads [features (20) | pyramid (4)]; the reference's order is pyramid first."""
    g = grads.pop(P.wfinal, None)
    if g is not None:
        grads[conv.weight] = torch.cat([g[:, 20:], g[:, :20]], dim=1).contiguous()


class _DensePyramid(_DenseBase):
    """tail of dehaze1113.Dense2 (:688-699) and dehaze22.Dense (:634-660): LeakyReLU(conv_refin) -> four-scale head (32 / 16 / 8 / 4)
    -> refine3(24 -> 3) -> tanh."""

    def __init__(self):
        super().__init__()
        self._make_layers()
        self.refine3 = nn.Conv2d(20 + 4, 3, kernel_size=3, stride=1, padding=1)
        self.upsample = nn.functional.interpolate
        self.relu = nn.LeakyReLU(0.2, inplace=True)

    def _build_tail(self, P, n, h, w, dev, train):
        P.head = E.new_act(n, h, w, 24, dev)          # [x9 (20) | pyramid (4)]; refine3's input channels permuted to match
        P.conv(E.View(P.cat8, 0, 19), P.weight(self.conv_refin.weight, 20, 19, 3), E.View(P.head, 0, 20), 3, pad=1, bias=self.conv_refin.bias,
               e_act=L.ACT_LEAKY02, label="conv_refin")
        P.pw = torch.zeros((4, 20), dtype=torch.float32, device=dev)
        P.pb = torch.zeros((4,), dtype=torch.float32, device=dev)
        for i, nm in enumerate(("conv1010", "conv1020", "conv1030", "conv1040")):
            conv = getattr(self, nm)
            P.copies += [(P.pw[i], conv.weight.view(20)), (P.pb[i:i + 1], conv.bias)]
        x20, y4 = E.View(P.head, 0, 20), E.View(P.head, 20, 4)
        P.op(lambda: E.pyramid_pool4(x20, P.pw, P.pb, 32, 0.2, y4),
             record=dict(kind="pyramid", src=x20, dst=y4, w=P.pw, b=P.pb, k0=32, slope=0.2,
                         sink=_pyramid_sink([getattr(self, nm) for nm in ("conv1010", "conv1020", "conv1030", "conv1040")])))
        P.wfinal = torch.zeros((3, 24, 3, 3), dtype=torch.float32, device=dev)
        P.wfinal.requires_grad_(self.refine3.weight.requires_grad)
        P.conv(E.View(P.head), P.weight(P.wfinal, 3, 24, 3), None, 3, pad=1, bias=self.refine3.bias, e_act=L.ACT_TANH,
               y_fd=E.nchw_f32_view(P.out), label="refine3")
        P.keep += [x20, y4]

    def _derived_grads(self, P, grads):
        super()._derived_grads(P, grads)
        _permuted_final_grad(P, grads, self.refine3)

    def _refresh_tail(self, P):
        wf = self.refine3.weight
        P.wfinal.requires_grad_(wf.requires_grad)
        if getattr(P.wfinal, "_src_version", None) != (wf._version, wf.data_ptr()):
            P.wfinal[:, :20].copy_(wf.detach()[:, 4:])       # reference order: [pyramid 0-3 | x9 4-23]
            P.wfinal[:, 20:].copy_(wf.detach()[:, :4])
            P.wfinal._src_version = (wf._version, wf.data_ptr())


class Dense(_DenseBase):
    """dehaze1113.py:431-570: tail conv_refin -> batchnorm20 -> LeakyReLU -> refine3(20 -> 3) -> tanh."""

    def __init__(self):
        super().__init__()
        self._make_layers()
        self.refine3 = nn.Conv2d(20, 3, kernel_size=3, stride=1, padding=1)
        self.upsample = nn.functional.interpolate
        self.relu = nn.LeakyReLU(0.2, inplace=True)
        self.batchnorm20 = nn.BatchNorm2d(20)
        self.batchnorm1 = nn.BatchNorm2d(1)            # registered, never called (:466)

    def _build_tail(self, P, n, h, w, dev, train):
        P.head = E.new_act(n, h, w, 24, dev, zero=True)
        s20 = ChanStats(24, dev)
        P.conv(E.View(P.cat8, 0, 19), P.weight(self.conv_refin.weight, 20, 19, 3), E.View(P.head, 0, 20), 3, pad=1, bias=self.conv_refin.bias,
               stats=s20 if train else None, label="conv_refin")
        P.conv(E.View(P.head, 0, 20), P.weight(self.refine3.weight, 3, 20, 3), None, 3, pad=1, bias=self.refine3.bias,
               pro=P.bn_prologue(self.batchnorm20, s20, n * h * w, act=L.ACT_LEAKY02), e_act=L.ACT_TANH, y_fd=E.nchw_f32_view(P.out), label="refine3")
        P.keep.append(s20)

    def _refresh_tail(self, P):
        pass


class Dense2(_DensePyramid):
    """dehaze1113.py:572-699."""


# ---------------------------------------------------------------------------------------
# Fusion-discriminator
# ---------------------------------------------------------------------------------------
class _Named(nn.Sequential):
    """Parameter container keeping the reference's nested key names
    (`main.layer2.layer2.conv.weight`)."""

    def __init__(self, **children):
        super().__init__()
        for k, v in children.items():
            self.add_module(k, v)

    def forward(self, *a, **k):
        raise RuntimeError("container only; the math runs in the enclosing network's HIP plan")


def blockUNet1(in_c, out_c, name, transposed=False, bn=False, relu=True, dropout=False):
    """dehaze1113.py:29-43 (3x3 stride-1 block).  Only the form `D` uses is supported on
    the HIP path: conv (not transposed), no dropout."""
    if transposed or dropout:
        raise NotImplementedError("blockUNet1(transposed/dropout) is unused by FD-GAN's D")
    kids = {"relu" if relu else "leakyrelu": nn.ReLU(inplace=True) if relu else nn.LeakyReLU(0.2, inplace=True),
            "conv": nn.Conv2d(in_c, out_c, 3, 1, 1, bias=False)}
    if bn:
        kids["bn"] = nn.BatchNorm2d(out_c)
    return _Named(**{name: _Named(**kids)})


class D(_PlannedModule):
    """dehaze1113.py:188-230.  (B,nc,H,W) -> (B,1,H/2-2,W/2-2) sigmoid map."""

    def __init__(self, nc, nf):
        super().__init__()
        self.nc, self.nf = nc, nf
        self.main = _Named(
            layer1=_Named(conv=nn.Conv2d(nc, nf, 4, 2, 1, bias=False)),
            layer2=blockUNet1(nf, nf * 2, "layer2", transposed=False, bn=True, relu=False, dropout=False),
            layer3=blockUNet1(nf * 2, nf * 4, "layer3", transposed=False, bn=True, relu=False, dropout=False),
            layer4=_Named(leakyrelu=nn.LeakyReLU(0.2, inplace=True), conv=nn.Conv2d(nf * 4, nf * 8, 4, 1, 1, bias=False)),
            layer5=_Named(leakyrelu=nn.LeakyReLU(0.2, inplace=True), conv=nn.Conv2d(nf * 8, 1, 4, 1, 1, bias=False),
                          sigmoid=nn.Sigmoid()))

    def _build_plan(self, shape, dev):
        n, c, h, w = shape
        if c != self.nc:
            raise ValueError("D expects %d input channels, got %d" % (self.nc, c))
        nf = self.nf
        r8 = lambda v: (v + 7) // 8 * 8
        P = NetPlan(dev)
        h1, w1 = (h + 2 - 4) // 2 + 1, (w + 2 - 4) // 2 + 1
        h4, w4 = h1 - 1, w1 - 1
        h5, w5 = h4 - 1, w4 - 1
        if h5 < 1 or w5 < 1:
            raise ValueError("input %dx%d too small for D" % (h, w))
        P.xin = E.new_act(n, h, w, r8(c), dev, zero=True)
        a1 = E.new_act(n, h1, w1, r8(nf), dev)
        a2 = E.new_act(n, h1, w1, r8(2 * nf), dev)
        a3 = E.new_act(n, h1, w1, r8(4 * nf), dev)
        a4 = E.new_act(n, h4, w4, r8(8 * nf), dev)
        P.out_shape = (n, 1, h5, w5)
        m = self.main
        l2, l3 = m.layer2.layer2, m.layer3.layer3
        s2, s3 = ChanStats(r8(2 * nf), dev), ChanStats(r8(4 * nf), dev)
        cnt = n * h1 * w1
        lrelu = E.make_prologue(act=L.ACT_LEAKY02)
        # views carry the 8-padded channel count: padded channels are stored as zeros
        P.conv(E.View(P.xin, 0, c), P.weight(m.layer1.conv.weight, nf, c, 4), E.View(a1), 4, pad=1, stride=2)
        P.conv(E.View(a1, 0, nf), P.weight(l2.conv.weight, 2 * nf, nf, 3), E.View(a2), 3, pad=1, pro=lrelu,
               stats=s2 if l2.bn.training else None)
        P.conv(E.View(a2, 0, 2 * nf), P.weight(l3.conv.weight, 4 * nf, 2 * nf, 3), E.View(a3), 3, pad=1,
               pro=P.bn_prologue(l2.bn, s2, cnt, act=L.ACT_LEAKY02), stats=s3 if l3.bn.training else None)
        P.conv(E.View(a3, 0, 4 * nf), P.weight(m.layer4.conv.weight, 8 * nf, 4 * nf, 4), E.View(a4), 4, pad=1,
               pro=P.bn_prologue(l3.bn, s3, cnt, act=L.ACT_LEAKY02))
        P.a4 = a4
        P.acts, P.s2, P.s3 = (a1, a2, a3, a4), s2, s3
        P.w_last = P.weight(m.layer5.conv.weight, 1, 8 * nf, 4)
        P.last_desc = E.conv_desc(4, 1, 1, L.ACT_SIGMOID, False, cout=1)
        P.keep += [s2, s3, a1, a2, a3]
        return P.finish()

    def _forward_plan(self, x):
        """Runs the recorded forward; returns (plan, sigmoid map).  No autograd."""
        P = self._plan_for(x)
        with torch.no_grad():
            E.to_nhwc(x.detach().float().contiguous(), E.View(P.xin))
            P.launch()
            out = torch.empty(P.out_shape, dtype=torch.float32, device=x.device)
            E.conv2d(E.View(P.a4, 0, 8 * self.nf).fd, P.w_last, None, E.make_prologue(act=L.ACT_LEAKY02),
                     E.nchw_f32_view(out), P.last_desc)
        return P, out

    def forward(self, x):
        if _wants_grad(self, x):
            return _apply_plan_function(self, x)
        return self._forward_plan(x)[1]

    def _autograd_forward(self, x):
        P, out = self._forward_plan(x)
        return out, (P, out, bool(x.requires_grad))

    def _autograd_backward(self, state, dout):
        """dehaze1113.py:188-230 under autograd (train-mode BatchNorm): the recorded plan walked in reverse
        (fdgan_hip/backward.py); the gradient w.r.t. the 9-channel input -- what the generator's adversarial loss
        needs -- comes from the any-stride direct kernel (layer1 is 4x4 stride 2)."""
        P, out, need_dx = state
        B = _plan_backward(P)
        B.zero_()
        n, _, h5, w5 = out.shape
        g8 = B.persistent("g8", lambda: E.new_grad(n, h5, w5, 8, out.device))    # read by the (recorded) walk's first launches
        E.out_act_bwd(dout, out, L.ACT_SIGMOID, E.View(g8))
        grads = {}
        last = B.persistent("last", lambda: dict(x=E.View(P.a4, 0, 8 * self.nf), w=P.w_last, k=4, pad=1, stride=1, bias=None,
                                                 pro=E.make_prologue(act=L.ACT_LEAKY02)))
        B.run(grads, skip_dx_of={P.xin.data_ptr()}, head=(last, E.View(g8, 0, 1)))
        dx = None
        if need_dx:
            dx = torch.empty((n, self.nc, P.xin.shape[1], P.xin.shape[2]), dtype=torch.float32, device=out.device)
            E.conv_bwd_data_direct(B.G(E.View(P.acts[0], 0, self.nf)).fd, self.main.layer1.conv.weight.detach().contiguous(),
                                   E.conv_desc(4, 2, 1, cout=self.nf), dx)
        return dx, grads
