"""`models.dehaze22` surface of the reference (/root/reference/models/dehaze22.py): the pix2pix
PatchGAN discriminator `D(nc, nf)` (:114-156) on the HIP path.  The legacy DCPDN generators in that
file (G :205, G2 :364, dehaze :662) are not on FD-GAN's hot path and are not
provided; asking for them raises.

Dataflow (one recorded plan, NHWC bf16, BatchNorm folded into the consumer's prologue):
    x -> 4x4 s2 -> [LReLU | 4x4 s2 | BN] x2 -> LReLU, 4x4 s1, BN -> LReLU, 4x4 s1 (->1), sigmoid
"""
import torch
import torch.nn as nn

from fdgan_hip import engine as E
from fdgan_hip import lib as L
from fdgan_hip.netplan import ChanStats, NetPlan
from models.dehaze1113 import _Named, _PlannedModule, _apply_plan_function, _plan_backward, _wants_grad


def blockUNet(in_c, out_c, name, transposed=False, bn=False, relu=True, dropout=False):
    """dehaze22.py:51-65 (4x4 stride-2 block); only the forward-conv form `D` uses runs on the HIP path."""
    if transposed or dropout:
        raise NotImplementedError("blockUNet(transposed/dropout) belongs to the legacy DCPDN nets, not to FD-GAN")
    kids = {"relu" if relu else "leakyrelu": nn.ReLU(inplace=True) if relu else nn.LeakyReLU(0.2, inplace=True),
            "conv": nn.Conv2d(in_c, out_c, 4, 2, 1, bias=False)}
    if bn:
        kids["bn"] = nn.BatchNorm2d(out_c)
    return _Named(**{name: _Named(**kids)})


class D(_PlannedModule):
    """dehaze22.py:114-156.  (B,nc,H,W) -> (B,1,H/8-2,W/8-2) sigmoid patch map (30x30 at 256^2)."""

    def __init__(self, nc, nf):
        super().__init__()
        self.nc, self.nf = nc, nf
        self.main = _Named(
            layer1=_Named(conv=nn.Conv2d(nc, nf, 4, 2, 1, bias=False)),
            layer2=blockUNet(nf, nf * 2, "layer2", transposed=False, bn=True, relu=False, dropout=False),
            layer3=blockUNet(nf * 2, nf * 4, "layer3", transposed=False, bn=True, relu=False, dropout=False),
            layer4=_Named(leakyrelu=nn.LeakyReLU(0.2, inplace=True), conv=nn.Conv2d(nf * 4, nf * 8, 4, 1, 1, bias=False),
                          bn=nn.BatchNorm2d(nf * 8)),
            layer5=_Named(leakyrelu=nn.LeakyReLU(0.2, inplace=True), conv=nn.Conv2d(nf * 8, 1, 4, 1, 1, bias=False),
                          sigmoid=nn.Sigmoid()))

    def _build_plan(self, shape, dev):
        n, c, h, w = shape
        if c != self.nc:
            raise ValueError("D expects %d input channels, got %d" % (self.nc, c))
        nf = self.nf
        r8 = lambda v: (v + 7) // 8 * 8
        half = lambda v: (v + 2 - 4) // 2 + 1
        hs, ws = [h], [w]
        for _ in range(3):
            hs.append(half(hs[-1])), ws.append(half(ws[-1]))
        hs += [hs[3] - 1, hs[3] - 2]
        ws += [ws[3] - 1, ws[3] - 2]
        if hs[5] < 1 or ws[5] < 1:
            raise ValueError("input %dx%d too small for dehaze22.D" % (h, w))
        P = NetPlan(dev)
        P.xin = E.new_act(n, h, w, r8(c), dev, zero=True)
        a1 = E.new_act(n, hs[1], ws[1], r8(nf), dev)
        a2 = E.new_act(n, hs[2], ws[2], r8(2 * nf), dev)
        a3 = E.new_act(n, hs[3], ws[3], r8(4 * nf), dev)
        a4 = E.new_act(n, hs[4], ws[4], r8(8 * nf), dev)
        P.out_shape = (n, 1, hs[5], ws[5])
        m = self.main
        l2, l3, l4 = m.layer2.layer2, m.layer3.layer3, m.layer4
        s2, s3, s4 = ChanStats(r8(2 * nf), dev), ChanStats(r8(4 * nf), dev), ChanStats(r8(8 * nf), dev)
        lrelu = E.make_prologue(act=L.ACT_LEAKY02)
        P.conv(E.View(P.xin, 0, c), P.weight(m.layer1.conv.weight, nf, c, 4), E.View(a1), 4, pad=1, stride=2)
        P.conv(E.View(a1, 0, nf), P.weight(l2.conv.weight, 2 * nf, nf, 4), E.View(a2), 4, pad=1, stride=2, pro=lrelu,
               stats=s2 if l2.bn.training else None)
        P.conv(E.View(a2, 0, 2 * nf), P.weight(l3.conv.weight, 4 * nf, 2 * nf, 4), E.View(a3), 4, pad=1, stride=2,
               pro=P.bn_prologue(l2.bn, s2, n * hs[2] * ws[2], act=L.ACT_LEAKY02), stats=s3 if l3.bn.training else None)
        P.conv(E.View(a3, 0, 4 * nf), P.weight(l4.conv.weight, 8 * nf, 4 * nf, 4), E.View(a4), 4, pad=1,
               pro=P.bn_prologue(l3.bn, s3, n * hs[3] * ws[3], act=L.ACT_LEAKY02), stats=s4 if l4.bn.training else None)
        P.a4, P.a1 = a4, a1
        P.w_last = P.weight(m.layer5.conv.weight, 1, 8 * nf, 4)
        P.last_desc = E.conv_desc(4, 1, 1, L.ACT_SIGMOID, False, cout=1)
        P.last_pro = P.bn_prologue(l4.bn, s4, n * hs[4] * ws[4], act=L.ACT_LEAKY02)
        P.keep += [s2, s3, s4, a1, a2, a3]
        return P.finish()

    def _run(self, x):
        P = self._plan_for(x)
        with torch.no_grad():
            E.to_nhwc(x.detach().float().contiguous(), E.View(P.xin))
            P.launch()
            out = torch.empty(P.out_shape, dtype=torch.float32, device=x.device)
            E.conv2d(E.View(P.a4, 0, 8 * self.nf).fd, P.w_last, None, P.last_pro, E.nchw_f32_view(out), P.last_desc)
        return P, out

    def forward(self, x):
        if _wants_grad(self, x):
            return _apply_plan_function(self, x)
        return self._run(x)[1]

    def _autograd_forward(self, x):
        P, out = self._run(x)
        return out, (P, out, bool(x.requires_grad))

    def _autograd_backward(self, state, dout):
        """dehaze22.py:114-156 under autograd (train-mode BatchNorm).  The 4x4 stride-2 data gradients use the
        any-stride direct kernel."""
        P, out, need_dx = state
        if not self.main.layer4.bn.training:
            raise NotImplementedError("dehaze22.D backward is built for train-mode BatchNorm")
        B = _plan_backward(P)
        B.zero_()
        n, _, h5, w5 = out.shape
        g8 = E.new_act(n, h5, w5, 8, out.device)
        E.out_act_bwd(dout, out, L.ACT_SIGMOID, E.View(g8))
        grads = {}
        last = dict(x=E.View(P.a4, 0, 8 * self.nf), w=P.w_last, k=4, pad=1, stride=1, bias=None,
                    pro=E.prologue_without_side_effects(P.last_pro))
        B.conv_backward(last, E.View(g8, 0, 1), grads)
        B.run(grads, skip_dx_of={P.xin.data_ptr()})
        dx = None
        if need_dx:
            dx = torch.empty((n, self.nc, P.xin.shape[1], P.xin.shape[2]), dtype=torch.float32, device=out.device)
            E.conv_bwd_data_direct(B.G(E.View(P.a1, 0, self.nf)).fd, self.main.layer1.conv.weight.detach().contiguous(),
                                   E.conv_desc(4, 2, 1, cout=self.nf), dx)
        return dx, grads


def _legacy(name):
    def ctor(*a, **k):
        raise NotImplementedError("models.dehaze22.%s is a legacy DCPDN network outside FD-GAN's hot path "
                                  "(reference demo.py uses models.dehaze1113.FDGAN)" % name)
    ctor.__name__ = name
    return ctor


G, G2, dehaze, D_tran = _legacy("G"), _legacy("G2"), _legacy("dehaze"), _legacy("D_tran")
