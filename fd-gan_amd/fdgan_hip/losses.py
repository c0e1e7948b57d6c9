"""Scalar losses on the HIP path: what a training loop over the reference's networks puts between the
network outputs and `backward()`.

    l1_loss / mse_loss / bce_loss   F.l1_loss / F.mse_loss / F.binary_cross_entropy (mean reduction) on fp32 tensors:
                                    value and gradient from one HIP pass (fdgan_loss_f32), ordered reduction.
    vgg_perceptual                  sum_k F.mse_loss(Vgg16_k(x), Vgg16_k(target)) over the four tapped feature maps
                                    (/root/reference/myutils/vgg16.py:27-49), computed on the NHWC bf16 buffers the
                                    convolutions wrote: no NCHW fp32 copies, the gradient lands in the plan's gradient
                                    buffers directly.

Every function returns a 0-dim device tensor and never synchronises with the host.
"""
import ctypes as C

import torch

from . import engine as E
from . import lib as L

_KINDS = {"l1": 0, "mse": 1, "bce": 2}
_PARTIAL = {}


def _partial(dev, n=8192):
    """Per-(device, stream) scratch for the per-workgroup sums (a launch writes at most 2048): losses evaluated on
    different streams (train.py runs the discriminator's real half beside the generator's forward) must not share it."""
    key = (dev.index, n, torch.cuda.current_stream(dev).cuda_stream)
    if key not in _PARTIAL:
        _PARTIAL[key] = torch.empty(n, dtype=torch.float32, device=dev)
    return _PARTIAL[key]


def _loss_fwd(kind, x, t, want_grad):
    E.require_gpu(x, kind + "_loss")
    xf = x.detach().float().contiguous()
    tf, tc = None, 0.0
    if isinstance(t, torch.Tensor):
        if t.shape != x.shape:
            raise ValueError("%s_loss: shapes differ: %s vs %s" % (kind, tuple(x.shape), tuple(t.shape)))
        tf = t.detach().float().contiguous()
    else:
        tc = float(t)
    grad = torch.empty_like(xf) if want_grad else None
    part = _partial(xf.device)
    out = torch.empty((), dtype=torch.float32, device=xf.device)
    nparts = C.c_int64(0)
    lib = L.load()
    L.check(lib.fdgan_loss_f32(_KINDS[kind], xf.data_ptr(), tf.data_ptr() if tf is not None else None, tc, xf.numel(),
                               grad.data_ptr() if grad is not None else None, part.data_ptr(), part.numel(),
                               C.byref(nparts), E.stream_ptr()), "loss_f32")
    L.check(lib.fdgan_sum_partials(part.data_ptr(), nparts.value, 1.0 / xf.numel(), out.data_ptr(), E.stream_ptr()),
            "sum_partials")
    return out, grad


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, kind, x, t):
        out, grad = _loss_fwd(kind, x, t, True)
        ctx.save_for_backward(grad)
        ctx.dtype = x.dtype
        return out

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return None, (grad * g).to(ctx.dtype), None       # g stays on the device: no sync


def _loss(kind, x, t):
    if isinstance(t, torch.Tensor) and t.requires_grad and torch.is_grad_enabled():
        raise NotImplementedError("%s_loss is differentiable w.r.t. its first argument only" % kind)
    if torch.is_grad_enabled() and x.requires_grad:
        return _LossFn.apply(kind, x, t)
    return _loss_fwd(kind, x, t, False)[0]


def l1_loss(x, target):
    """F.l1_loss(x, target) (mean)."""
    return _loss("l1", x, target)


def mse_loss(x, target):
    """F.mse_loss(x, target) (mean)."""
    return _loss("mse", x, target)


def bce_loss(p, target):
    """F.binary_cross_entropy(p, target) (mean); `target` a tensor like p or a python number (1.0 real / 0.0 fake)."""
    return _loss("bce", p, target)


# ---- perceptual loss ----------------------------------------------------------------------------------------------
def _mse_taps_fwd(Px, Pt):
    part = _partial(Px.device)
    out = torch.empty((), dtype=torch.float32, device=Px.device)
    lib = L.load()
    off = 0
    for a, b in zip(Px.taps, Pt.taps):
        n, h, w, c = a.shape
        np_ = C.c_int64(0)
        L.check(lib.fdgan_mse_nhwc_fwd(C.byref(a.fd), C.byref(b.fd), 1.0 / (n * h * w * c), part.data_ptr() + 4 * off,
                                       part.numel() - off, C.byref(np_), E.stream_ptr()), "mse_nhwc_fwd")
        off += np_.value
    L.check(lib.fdgan_sum_partials(part.data_ptr(), off, 1.0, out.data_ptr(), E.stream_ptr()), "sum_partials")
    return out


class _PerceptualFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, vgg, x, target):
        Pt = target if not isinstance(target, torch.Tensor) else vgg.run_nhwc(target, slot=1)   # targets live in their own plan
        Px = vgg.run_nhwc(x, slot=0)
        ctx.vgg, ctx.Px, ctx.Pt, ctx.gen = vgg, Px, Pt, (Px._gen, Pt._gen)
        ctx.dtype = x.dtype
        return _mse_taps_fwd(Px, Pt)

    @staticmethod
    def backward(ctx, g):
        Px, Pt = ctx.Px, ctx.Pt
        if (Px._gen, Pt._gen) != ctx.gen:
            raise RuntimeError("vgg_perceptual: the Vgg16 plan ran again between this forward and its backward; its "
                               "activations were overwritten (call backward first, or use another Vgg16 instance)")
        B = ctx.vgg.plan_backward(Px)
        B.enable_relu_premask()        # gradients of ReLU outputs are masked where they are produced: no separate mask passes
        B.zero_()
        gs = g.detach().float().contiguous()
        lib = L.load()
        for a, b in zip(Px.taps, Pt.taps):
            n, h, w, c = a.shape
            L.check(lib.fdgan_mse_nhwc_bwd(C.byref(a.fd), C.byref(b.fd), gs.data_ptr(), 2.0 / (n * h * w * c), 1,
                                           C.byref(B.G(a).fd), E.stream_ptr()), "mse_nhwc_bwd")
        B.run({}, skip_dx_of=())
        xin = Px.xin
        dx = torch.empty((xin.shape[0], 3, xin.shape[1], xin.shape[2]), dtype=torch.float32, device=xin.device)
        E.to_nchw(B.G(E.View(xin, 0, 3)), dx)
        return None, dx.to(ctx.dtype), None


def vgg_targets(vgg, target):
    """Feature maps of `target` in the module's target slot, for a later vgg_perceptual(vgg, x, <this>): lets a training
    step compute them early (on another stream, beside the generator's forward)."""
    E.require_gpu(target, "vgg_targets")
    with torch.no_grad():
        return vgg.run_nhwc(target, slot=1)


def vgg_perceptual(vgg, x, target):
    """sum over Vgg16's four taps of F.mse_loss(feature(x), feature(target)); differentiable w.r.t. x.  Vgg16's own
    parameters are treated as frozen (the reference loads a pretrained, fixed VGG16: myutils/utils.py:84-94).
    `target`: an image batch, or the result of vgg_targets(vgg, images)."""
    E.require_gpu(x, "vgg_perceptual")
    if isinstance(target, torch.Tensor):
        E.require_gpu(target, "vgg_perceptual")
    if any(p.requires_grad for p in vgg.parameters()):
        raise NotImplementedError("vgg_perceptual treats Vgg16 as a frozen feature extractor: call "
                                  "`for p in vgg.parameters(): p.requires_grad_(False)` (or use vgg(x) + mse_loss)")
    if torch.is_grad_enabled() and x.requires_grad:
        return _PerceptualFn.apply(vgg, x, target)
    with torch.no_grad():
        Pt = target if not isinstance(target, torch.Tensor) else vgg.run_nhwc(target, slot=1)
        Px = vgg.run_nhwc(x, slot=0)
        return _mse_taps_fwd(Px, Pt)
