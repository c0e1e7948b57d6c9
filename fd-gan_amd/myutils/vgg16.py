"""MI355X-native `myutils.vgg16.Vgg16`: the perceptual-loss feature extractor
(/root/reference/myutils/vgg16.py:6-49).  Same constructor, same 26 state_dict tensors
(conv5_x registered but never called, :23-25), same output: [relu1_2, relu2_2, relu3_3, relu4_3]
as NCHW fp32 tensors.  The ten 3x3 convs (+bias, +ReLU) run as NHWC bf16 implicit-GEMM MFMA
kernels, the three 2x2 max-pools as an NHWC kernel, all recorded in one plan per input shape.
"""
import torch
import torch.nn as nn

from fdgan_hip import engine as E
from fdgan_hip import lib as L
from fdgan_hip.backward import autograd_grads
from fdgan_hip.netplan import NetPlan

_CFG = [("conv1_1", 3, 64), ("conv1_2", 64, 64), "tap", "pool",
        ("conv2_1", 64, 128), ("conv2_2", 128, 128), "tap", "pool",
        ("conv3_1", 128, 256), ("conv3_2", 256, 256), ("conv3_3", 256, 256), "tap", "pool",
        ("conv4_1", 256, 512), ("conv4_2", 512, 512), ("conv4_3", 512, 512), "tap"]
_UNUSED = [("conv5_1", 512, 512), ("conv5_2", 512, 512), ("conv5_3", 512, 512)]


class Vgg16(torch.nn.Module):
    def __init__(self):
        super(Vgg16, self).__init__()
        for item in _CFG + _UNUSED:
            if isinstance(item, tuple):
                setattr(self, item[0], nn.Conv2d(item[1], item[2], kernel_size=3, stride=1, padding=1))

    def release_plans(self):
        """Closes and drops every cached plan (models.dehaze1113._PlannedModule.release_plans)."""
        for pl in self.__dict__.pop("_plans", {}).values():
            pl.close()

    def _apply(self, fn, *a, **k):
        self.release_plans()
        return super()._apply(fn, *a, **k)

    def _plan_for(self, X, slot=0):
        E.require_gpu(X, "Vgg16.forward")
        if X.dim() != 4 or X.shape[1] != 3:
            raise ValueError("Vgg16 expects a Bx3xHxW tensor, got %s" % (tuple(X.shape),))
        cache = self.__dict__.setdefault("_plans", {})
        key = (tuple(X.shape), X.device.index, slot)
        P = cache.get(key)
        if P is not None and P.param_ptrs() != P._built_ptrs:
            cache.pop(key).close()
            P = None
        if P is None:
            P = self._build(tuple(X.shape), X.device)
            P._built_ptrs = P.param_ptrs()
            cache[key] = P
        return P

    def _build(self, shape, dev):
        n, _, h, w = shape
        if h % 8 or w % 8:
            raise ValueError("Vgg16 on the HIP path needs H and W to be multiples of 8, got %dx%d" % (h, w))
        P = NetPlan(dev)
        P.xin = E.new_act(n, h, w, 8, dev, zero=True)
        cur, cur_c = P.xin, 3
        P.taps = []
        P._gen = 0                 # bumped by every run: a backward checks that its activations are still the ones it saw
        for item in _CFG:
            if item == "tap":
                P.taps.append(E.View(cur, 0, cur_c))
            elif item == "pool":
                nxt = E.new_act(n, cur.shape[1] // 2, cur.shape[2] // 2, cur_c, dev)
                src, dst = E.View(cur, 0, cur_c), E.View(nxt)
                P.maxpool(src, dst)
                cur = nxt
            else:
                name, cin, cout = item
                conv = getattr(self, name)
                nxt = E.new_act(n, cur.shape[1], cur.shape[2], cout, dev)
                P.conv(E.View(cur, 0, cin), P.weight(conv.weight, cout, cin, 3), E.View(nxt), 3, pad=1, bias=conv.bias,
                       e_act=L.ACT_RELU, label=name)
                cur, cur_c = nxt, cout
        return P.finish()

    def run_nhwc(self, X, slot=0):
        """Runs the plan of `slot` on X and returns it: the four feature maps are the NHWC bf16 views `plan.taps`, valid
        until the same slot runs again (fdgan_hip.losses.vgg_perceptual keeps inputs and targets in two slots)."""
        P = self._plan_for(X, slot)
        with torch.no_grad():
            E.to_nhwc(X.detach().float().contiguous(), E.View(P.xin))
            P.launch()
        P._gen += 1
        return P

    def plan_backward(self, P):
        from fdgan_hip.backward import PlanBackward
        if getattr(P, "_bwd", None) is None:
            P._bwd = PlanBackward(P)
        return P._bwd

    def _run(self, X):
        P = self.run_nhwc(X)
        with torch.no_grad():
            outs = []
            for v in P.taps:
                nn_, hh, ww, cc = v.shape
                o = torch.empty((nn_, cc, hh, ww), dtype=torch.float32, device=X.device)
                E.to_nchw(v, o)
                outs.append(o)
        return P, outs

    def forward(self, X):
        if torch.is_grad_enabled() and (X.requires_grad or any(p.requires_grad for p in self.parameters())):
            params = tuple(p for p in self.parameters() if p.requires_grad)
            return list(_VggFunction.apply(self, X, *params))
        return self._run(X)[1]

    def _backward(self, P, douts, need_dx):
        """Perceptual-loss path: gradients of the four tapped feature maps back to the input image (and to the
        filters, if they are not frozen)."""
        B = self.plan_backward(P)
        B.zero_()
        for v, d in zip(P.taps, douts):
            if d is not None:
                E.to_nhwc(d.detach().float().contiguous(), B.G(v))
        grads = {}
        B.run(grads, skip_dx_of=() if need_dx else {P.xin.data_ptr()})
        dx = None
        if need_dx:
            dx = torch.empty((P.xin.shape[0], 3, P.xin.shape[1], P.xin.shape[2]), dtype=torch.float32, device=P.xin.device)
            E.to_nchw(B.G(E.View(P.xin, 0, 3)), dx)
        return dx, grads


class _VggFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, X, *params):
        P, outs = module._run(X)
        ctx.module, ctx.plan, ctx.params, ctx.need_dx, ctx.gen = module, P, params, X.requires_grad, P._gen
        return tuple(outs)

    @staticmethod
    def backward(ctx, *douts):
        if ctx.plan._gen != ctx.gen:       # another forward of the same shape overwrote the activations this backward reads
            raise RuntimeError("Vgg16: the module ran again on an input of the same shape between this forward and its "
                               "backward; the plan's activations were overwritten.  Compute targets first (under "
                               "torch.no_grad()), or use fdgan_hip.losses.vgg_perceptual, which keeps them apart.")
        dx, grads = ctx.module._backward(ctx.plan, douts, ctx.need_dx)
        return (None, dx) + autograd_grads(grads, ctx.params)
