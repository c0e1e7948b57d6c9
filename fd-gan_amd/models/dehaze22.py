"""`models.dehaze22` surface of the reference (/root/reference/models/dehaze22.py): the pix2pix
PatchGAN discriminator `D(nc, nf)` (:114-156, `D_tran` :159-201 is the same network) and, from the
DCPDN-era generators of that file (not on FD-GAN's hot path), the two U-Nets
`G` (:205-362) and `G2` (:364-488) -- forward and (train mode) backward, on the same primitives: 4x4 stride-2 convolutions,
ConvTranspose2d(4, 2, 1) as four stride-1 3x3 convolutions (one per output parity, strided output
views), train- or eval-mode BatchNorm folded into the consumers' prologues, train-mode Dropout2d,
the multi-scale pooling head as one kernel -- and `Dense` (:531-660, built in models/dehaze1113.py).
`dehaze` (:662-753) composes them around `fdgan_scatter_dehaze`.

Dataflow (one recorded plan, NHWC bf16, BatchNorm folded into the consumer's prologue):
    x -> 4x4 s2 -> [LReLU | 4x4 s2 | BN] x2 -> LReLU, 4x4 s1, BN -> LReLU, 4x4 s1 (->1), sigmoid
"""
import torch
import torch.nn as nn

from fdgan_hip import engine as E
from fdgan_hip import lib as L
from fdgan_hip.netplan import ChanStats, NetPlan
from fdgan_hip.backward import autograd_grads, grad_target
from models.dehaze1113 import (BottleneckBlock, TransitionBlock, _DensePyramid, _Named, _PlannedModule, _apply_plan_function,
                               _bump_generation, _check_generation, _permuted_final_grad, _plan_backward, _pyramid_sink, _wants_grad)


def blockUNet(in_c, out_c, name, transposed=False, bn=False, relu=True, dropout=False):
    """dehaze22.py:51-65: [ReLU | LeakyReLU(0.2)] -> Conv2d(4, 2, 1) | ConvTranspose2d(4, 2, 1) -> [BatchNorm2d] -> [Dropout2d(0.5)],
    as a parameter container with the reference's key names (`layer2.layer2.conv.weight`, `dlayer7.dlayer7.tconv.weight`)."""
    kids = {"relu" if relu else "leakyrelu": nn.ReLU(inplace=True) if relu else nn.LeakyReLU(0.2, inplace=True)}
    if not transposed:
        kids["conv"] = nn.Conv2d(in_c, out_c, 4, 2, 1, bias=False)
    else:
        kids["tconv"] = nn.ConvTranspose2d(in_c, out_c, 4, 2, 1, bias=False)
    if bn:
        kids["bn"] = nn.BatchNorm2d(out_c)
    if dropout:
        kids["dropout"] = nn.Dropout2d(0.5, inplace=True)
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
        B = _plan_backward(P)
        B.zero_()
        n, _, h5, w5 = out.shape
        g8 = E.new_grad(n, h5, w5, 8, out.device)
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


class D_tran(D):
    """dehaze22.py:159-201: layer for layer the network of `D` (same children, same key names)."""


# ConvTranspose2d(4, 2, 1): out[2 m + a] = sum_i x[i] w[2 (m - i) + a + 1].  Per output parity a that is a 3-tap correlation
# over x[m - 1 .. m + 1] with taps (w[3], w[1], 0) for a = 0 and (0, w[2], w[0]) for a = 1 -- in 2-D four 3x3 stride-1 pad-1
# convolutions whose results interleave.  (5 of the 9 taps are zero: the legacy nets pay 2.25x the MFMA work of a dedicated
# 2x2 kernel and get every forward kernel of the hot path in exchange.)
_TAP = {0: (3, 1, None), 1: (None, 2, 0)}


def _phase_filters(w_t, out):
    """w_t: ConvTranspose2d weight (cin, cout, 4, 4) -> out: four (cout, cin, 3, 3) tensors, the conv filters of parities (a, b)."""
    wt = w_t.detach().permute(1, 0, 2, 3)
    for a in range(2):
        for b in range(2):
            f = out[a * 2 + b]
            f.zero_()
            for ty, ky in enumerate(_TAP[a]):
                for tx, kx in enumerate(_TAP[b]):
                    if ky is not None and kx is not None:
                        f[:, :, ty, tx] = wt[:, :, ky, kx]
    return out


def _phase_filter_grads(dfilt, w_t):
    """Inverse of _phase_filters for gradients: four (cout, cin, 3, 3) -> (cin, cout, 4, 4) (every tap of the transposed filter sits in
    exactly one parity filter); a missing entry of dfilt counts as zero."""
    g = torch.zeros_like(w_t)
    for a in range(2):
        for b in range(2):
            d = dfilt[a * 2 + b]
            if d is None:
                continue
            for ty, ky in enumerate(_TAP[a]):
                for tx, kx in enumerate(_TAP[b]):
                    if ky is not None and kx is not None:
                        g[:, :, ky, kx] = d[:, :, ty, tx].permute(1, 0)
    return g


class _BnTable:
    """What backward.py reads off a BatchNorm module, for a per-channel table that holds two norms side by side: `weight` / `bias`
    are the table's gamma / beta rows (leaves that receive the table-wide dgamma / dbeta; `_UNet._table_grads` hands the halves to
    the modules they came from)."""

    def __init__(self, gamma, beta):
        self.weight, self.bias = gamma, beta


class _Stats:
    """(mean, var) tensor pair: what NetPlan.bn_prologue / NetPlan.conv(stats=...) take; here slices of a concat buffer's."""

    def __init__(self, mean, var):
        self.mean, self.var = mean, var


class _UNet(_PlannedModule):
    """The 8-level pix2pix U-Net shared by `G` and `G2` (dehaze22.py:205-362, :364-488); `loss.backward()` works in train mode
    (`_autograd_backward`).

    Buffers (NHWC bf16): level k = 1..7 owns ONE concat buffer cat_k = [dout_{k+1} | out_k] at H / 2^k -- the encoder conv of
    level k stores out_k into its right half, the decoder's transposed conv of level k+1 stores dout_{k+1} into the left
    half, so `torch.cat` (dehaze22.py:320-333) never runs.  Buffers hold RAW conv outputs; BatchNorm and the in-place
    (Leaky)ReLU of the consumer are its prologue.  The reference's in-place activations (dehaze22.py:54-56) make the skip
    tensors leaky_relu(out_k); the decoder then applies ReLU to the concatenation, and relu(leaky_relu(v)) = relu(v): one
    per-channel prologue (batch or running statistics of two different norms side by side, then ReLU) covers both halves.
    Train-mode Dropout2d (dlayers 8, 7, 6: at most 8 x 8 pixels) is applied together with that layer's BatchNorm by one small
    pass over the stored tensor, with an (N, C) mask drawn by torch's generator exactly as F.dropout2d draws it."""

    def __init__(self, input_nc, output_nc, nf):
        super().__init__()
        self.input_nc, self.output_nc, self.nf = input_nc, output_nc, nf
        self.ce = [nf, nf * 2, nf * 4, nf * 8, nf * 8, nf * 8, nf * 8, nf * 8]          # out_1 .. out_8
        self.cd = {8: nf * 8, 7: nf * 8, 6: nf * 8, 5: nf * 8, 4: nf * 4, 3: nf * 2, 2: nf}   # dout_8 .. dout_2
        if nf % 8:
            raise ValueError("the HIP path needs nf to be a multiple of 8, got %d" % nf)

    def _make_layers(self, last_cout):
        nf, ce = self.nf, self.ce
        self.layer1 = _Named(layer1=nn.Conv2d(self.input_nc, nf, 4, 2, 1, bias=False))
        for k in range(2, 9):
            setattr(self, "layer%d" % k, blockUNet(ce[k - 2], ce[k - 1], "layer%d" % k, transposed=False, bn=True, relu=False))
        self.dlayer8 = blockUNet(nf * 8, nf * 8, "dlayer8", transposed=True, bn=False, relu=True, dropout=True)
        for k in range(7, 1, -1):
            setattr(self, "dlayer%d" % k, blockUNet(self.cd[k + 1] + ce[k - 1], self.cd[k], "dlayer%d" % k, transposed=True, bn=True,
                                                     relu=True, dropout=k >= 6))
        self.dlayer1 = _Named(dlayer1=_Named(relu=nn.ReLU(inplace=True), tconv=nn.ConvTranspose2d(nf * 2, last_cout, 4, 2, 1, bias=False)))

    # ---- plan ---------------------------------------------------------------------------------------------
    def _tconv(self, P, src, pro, w_param, cout, cin, dst_buf, c0, hin, win, e_act, stats, count, label):
        """ConvTranspose2d(4, 2, 1) of `src` (View, hin x win) into channels [c0, c0 + cout) of dst_buf (2 hin x 2 win)."""
        dev = dst_buf.device
        filt = [torch.zeros((cout, cin, 3, 3), dtype=torch.float32, device=dev).requires_grad_(w_param.requires_grad) for _ in range(4)]
        P.derived.append((w_param, filt))     # leaves: the reverse walk leaves their gradients under these keys (`_derived_grads`)
        ws4 = [P.weight(filt[i], cout, cin, 3) for i in range(4)]
        ys = [E.StridedView(dst_buf, c0, cout, a, b, 2, 2, hin, win) for a in range(2) for b in range(2)]
        desc = E.conv_desc(3, 1, 1, e_act, False, cout=cout, w_layout=ws4[0].layout)
        need, info = 0, None
        if stats is not None:
            info = E.conv_info(src.fd, ys[0].fd, cout, desc, pro)
            need = 4 * info.stats_rows * info.stats_cpad * 2
        rows = info.stats_rows if info is not None else 0
        per = rows * (info.stats_cpad if info is not None else 0) * 2

        pro_rest = E.prologue_without_side_effects(pro)     # a norm's running statistics move once, not once per parity

        def run():
            for i in range(4):
                E.conv2d(src.fd, ws4[i], None, pro if i == 0 else pro_rest, ys[i].fd, desc, P.ws[i * per:] if stats is not None else None)
            if stats is not None:
                info4 = L.FdConvInfo()
                info4.stats_rows, info4.stats_cpad = 4 * rows, info.stats_cpad
                E.bn_finalize(P.ws, info4, cout, count, stats.mean, stats.var, 0)
        n = dst_buf.shape[0]
        P._ops.append((run, need, dict(label=label, k=4, cin=cin, cout=cout, n=n, h_out=2 * hin, w_out=2 * win,
                                       flops=2.0 * n * (2 * hin) * (2 * win) * cout * cin * 4, flops_done=2.0 * n * hin * win * 4 * cout * cin * 9,
                                       bytes=n * hin * win * cin * 2 + n * 4 * hin * win * cout * 2)))
        for i in range(4):     # what the reverse walk sees: four stride-1 convolutions whose outputs interleave
            P.records.append(dict(kind="conv", x=src, w=ws4[i], y=ys[i], k=3, pad=1, stride=1, bias=None, pro=pro_rest, e_act=e_act,
                                  upsample=False, rerun=None))
        P.keep += [src, pro, pro_rest, ws4, ys, desc, filt]

    def _build_plan(self, shape, dev):
        n, c, h, w = shape
        if c != self.input_nc:
            raise ValueError("%s expects %d input channels, got %d" % (type(self).__name__, self.input_nc, c))
        if h % 256 or w % 256:
            raise ValueError("%s: the 8-level U-Net halves the image eight times; H and W must be multiples of 256, got %dx%d"
                             % (type(self).__name__, h, w))
        nf, ce, cd = self.nf, self.ce, self.cd
        train = self.training
        keep = self.__dict__.get("_plan_keep", False)      # this plan will be walked in reverse (models.dehaze1113._PlannedModule)
        if train and n * (h // 256) * (w // 256) < 2:
            raise ValueError("train-mode BatchNorm at the 1x1 bottleneck needs more than one value per channel (batch >= 2 at 256x256)")
        r8 = lambda v: (v + 7) // 8 * 8
        P = NetPlan(dev)
        P.derived = []
        P.xin = E.new_act(n, h, w, r8(c), dev, zero=True)
        hs = [h >> k for k in range(9)]
        ws_ = [w >> k for k in range(9)]
        # concat buffers of levels 1..7, the bottleneck, per-channel tables of the decoder prologues
        cat = {k: E.new_act(n, hs[k], ws_[k], cd[k + 1] + ce[k - 1], dev) for k in range(1, 8)}
        out8 = E.new_act(n, hs[8], ws_[8], ce[7], dev)
        f32 = lambda cnt, v: torch.full((cnt,), v, dtype=torch.float32, device=dev)
        tab = {k: dict(mean=f32(cd[k + 1] + ce[k - 1], 0.0), var=f32(cd[k + 1] + ce[k - 1], 1.0 - 1e-5),
                       gamma=f32(cd[k + 1] + ce[k - 1], 1.0), beta=f32(cd[k + 1] + ce[k - 1], 0.0)) for k in range(1, 8)}
        P.tab, P.cat, P.out8 = tab, cat, out8
        enc_bn = {k: getattr(self, "layer%d" % k)[0].bn for k in range(2, 9)}
        dec_bn = {k: getattr(self, "dlayer%d" % k)[0].bn for k in range(2, 8)}
        P.copies = []      # (dst table slice, source tensor): refreshed before every launch (parameters / running statistics)
        P.running = []     # (bn, mean slice, var slice, count): train-mode running-statistics updates done after the launch
        P.masks = []       # (mask tensor (N, C)): train-mode Dropout2d
        lrelu = E.make_prologue(act=L.ACT_LEAKY02)
        # ---- encoder
        enc_w = lambda k: (self.layer1.layer1 if k == 1 else getattr(self, "layer%d" % k)[0].conv).weight
        out_view = lambda k: E.View(cat[k], cd[k + 1], ce[k - 1]) if k <= 7 else E.View(out8)
        enc_stats = {}
        for k in range(1, 9):
            src = E.View(P.xin, 0, c) if k == 1 else out_view(k - 1)
            cin = c if k == 1 else ce[k - 2]
            if k == 1:
                pro = None
            elif k == 2:
                pro = lrelu
            else:
                pro = P.bn_prologue(enc_bn[k - 1], enc_stats[k - 1], n * hs[k - 1] * ws_[k - 1], act=L.ACT_LEAKY02)
            st = None
            if k >= 2:
                if k <= 7:      # statistics land in the concat buffer's table, where the decoder reads them too
                    st = _Stats(tab[k]["mean"][cd[k + 1]:], tab[k]["var"][cd[k + 1]:])
                else:
                    st = ChanStats(ce[7], dev)
                enc_stats[k] = st
                if k <= 7:
                    P.copies += [(tab[k]["gamma"][cd[k + 1]:], enc_bn[k].weight), (tab[k]["beta"][cd[k + 1]:], enc_bn[k].bias)]
                    if not train:
                        P.copies += [(tab[k]["mean"][cd[k + 1]:], enc_bn[k].running_mean), (tab[k]["var"][cd[k + 1]:], enc_bn[k].running_var)]
            P.conv(src, P.weight(enc_w(k), ce[k - 1], cin, 4, stride=2), out_view(k), 4, pad=1, stride=2, pro=pro,
                   stats=st if (k >= 2 and train) else None, label="layer%d" % k)
        # ---- decoder: dlayer k reads cat_k (k <= 7) / out8 (k = 8) and writes dout_k into the left half of cat_{k-1}
        for k in range(8, 1, -1):
            blk = getattr(self, "dlayer%d" % k)[0]
            if k == 8:
                src, cin = E.View(out8), ce[7]
                pro = P.bn_prologue(enc_bn[8], enc_stats[8], n * hs[8] * ws_[8], act=L.ACT_RELU)
            else:
                src, cin = E.View(cat[k]), cd[k + 1] + ce[k - 1]
                pro = self._table_prologue(P, k, train)
            dst, cout = cat[k - 1], cd[k]
            bn = dec_bn.get(k)
            drop = train and k >= 6
            left = _Stats(tab[k - 1]["mean"][:cout], tab[k - 1]["var"][:cout])
            st = left if (bn is not None and train and not drop) else (ChanStats(cout, dev) if (bn is not None and train) else None)
            # a plan that is trained through keeps the raw transposed-conv output of the dropout levels (BatchNorm's backward needs
            # it; at most 8 x 8 pixels); an inference plan finishes it in place
            raw = E.new_act(n, hs[k - 1], ws_[k - 1], cout, dev) if (drop and keep) else None
            self._tconv(P, src, pro, blk.tconv.weight, cout, cin, raw if raw is not None else dst, 0, hs[k], ws_[k], L.ACT_NONE, st,
                        n * hs[k - 1] * ws_[k - 1], "dlayer%d" % k)
            if bn is not None and train:
                P.running.append((bn, st.mean, st.var, n * hs[k - 1] * ws_[k - 1]))
            if drop:        # BatchNorm + Dropout2d applied in one pass: the consumer sees finished values (identity entries in its table)
                mask = torch.ones((n, cout), dtype=torch.float32, device=dev)
                P.masks.append(mask)
                v = E.View(dst, 0, cout)
                u = E.View(raw) if raw is not None else v
                rec = dict(kind="bn_dropout", src=u, dst=v, mean=st.mean if bn is not None else None, var=st.var if bn is not None else None,
                           gamma=bn.weight if bn is not None else None, eps=bn.eps if bn is not None else 0.0, mask=mask, bn=bn) if raw is not None else None
                if bn is not None:
                    P.op(lambda v=v, u=u, st=st, bn=bn, mask=mask: E.bn_dropout(u, st.mean, st.var, bn.weight, bn.bias, bn.eps, mask, v), record=rec)
                else:
                    P.op(lambda v=v, u=u, mask=mask: E.bn_dropout(u, None, None, None, None, 0.0, mask, v), record=rec)
                P.keep += [v, u, mask, st]
            elif bn is not None:
                P.copies += [(tab[k - 1]["gamma"][:cout], bn.weight), (tab[k - 1]["beta"][:cout], bn.bias)]
                if not train:
                    P.copies += [(tab[k - 1]["mean"][:cout], bn.running_mean), (tab[k - 1]["var"][:cout], bn.running_var)]
        P.last_pro = self._table_prologue(P, 1, train)
        self._build_head(P, n, h, w, dev)
        P.keep += [cat, out8, tab, enc_stats]
        return P.finish()

    def _table_prologue(self, P, k, train):
        """BatchNorm + ReLU over the concat buffer of level k: [decoder norm of dlayer k+1 | encoder norm of layer k] side by side
        (identity entries where the left half holds finished values: a dropout level, or the norm-free dlayer8 / level-1 encoder)."""
        t, cl = P.tab[k], self.cd[k + 1]
        pro = E.make_prologue(act=L.ACT_RELU, mean=t["mean"], var=t["var"], gamma=t["gamma"], beta=t["beta"], eps=1e-5)
        if train:
            ident = []
            if k + 1 >= 6 or k + 1 == 8:            # dlayer 8 / 7 / 6: BatchNorm (if any) + Dropout2d already applied
                ident.append((0, cl))
            if k == 1:                               # layer1 has no norm
                ident.append((cl, cl + self.ce[0]))
            for nm in ("gamma", "beta"):
                t[nm].requires_grad_(True)
            pro._meta.update(bn=_BnTable(t["gamma"], t["beta"]), batch_stats=True, identity=ident)
        elif self.__dict__.get("_plan_keep", False):
            # eval mode under autograd (round 6): every entry is a constant -- running statistics, or the identity (mean 0, var 1 - eps,
            # gamma 1, beta 0) where a half has no norm -- so the backward is dx = gamma * rstd * dpre with dgamma / dbeta formed as
            # always (backward.py: _constant_entries); the identity entries' dgamma / dbeta belong to no module (`_table_grads`)
            for nm in ("gamma", "beta"):
                t[nm].requires_grad_(True)
            pro._meta.update(bn=_BnTable(t["gamma"], t["beta"]), batch_stats=False, identity=())
        return pro

    def _table_grads(self, P, grads):
        """dgamma / dbeta of the side-by-side tables -> the BatchNorm modules the halves belong to."""
        for k in range(1, 8):
            t, cl = P.tab[k], self.cd[k + 1]
            dg, db = grads.pop(t["gamma"], None), grads.pop(t["beta"], None)
            if dg is None:
                continue
            halves = []
            if k + 1 <= (5 if self.training else 7):  # left half: the decoder norm of dlayer k+1 (train mode: no dropout there; eval: 2..7)
                halves.append((getattr(self, "dlayer%d" % (k + 1))[0].bn, 0, cl))
            if k >= 2:                               # right half: the encoder norm of layer k
                halves.append((getattr(self, "layer%d" % k)[0].bn, cl, cl + self.ce[k - 1]))
            for bn, lo, hi in halves:
                if bn.weight.requires_grad:
                    grad_target(grads, bn.weight).add_(dg[lo:hi])
                    grad_target(grads, bn.bias).add_(db[lo:hi])

    def _derived_grads(self, P, grads):
        for w_t, filt in P.derived:
            if not w_t.requires_grad:
                continue
            d = [grads.pop(f, None) for f in filt]
            if any(x is not None for x in d):
                grad_target(grads, w_t).add_(_phase_filter_grads(d, w_t))

    def _run(self, x):
        P = self._plan_for(x)
        with torch.no_grad():
            for w_t, filt in P.derived:
                ver = w_t._version
                for f in filt:
                    f.requires_grad_(w_t.requires_grad)
                if getattr(filt[0], "_src_version", None) != (ver, w_t.data_ptr()):
                    _phase_filters(w_t, filt)
                    filt[0]._src_version = (ver, w_t.data_ptr())
            for dst, src_t in P.copies:
                dst.copy_(src_t.detach())
            forced = self.__dict__.get("_forced_dropout_masks")     # tests: the masks the oracle used, order dlayer8, 7, 6
            for i, mask in enumerate(P.masks):     # F.dropout2d draws a (N, C, 1, 1) Bernoulli(0.5) / 0.5 field (dehaze22.py:62-63)
                if forced is not None:
                    mask.copy_(forced[i])
                else:
                    mask.copy_(torch.nn.functional.dropout2d(torch.ones_like(mask)[:, :, None, None], 0.5, True)[:, :, 0, 0])
            E.to_nhwc(x.detach().float().contiguous(), E.View(P.xin))
            P.launch()
            for bn, mean, var, count in P.running:     # decoder norms: their consumers' prologues carry two norms at once
                m = bn.momentum if bn.momentum is not None else 0.1
                bn.running_mean.mul_(1 - m).add_(mean, alpha=m)
                bn.running_var.mul_(1 - m).add_(var, alpha=m * count / max(count - 1, 1))
                bn.num_batches_tracked.add_(1)
            return self._finish(P, x)

    def forward(self, x):
        if _wants_grad(self, x):
            return _apply_plan_function(self, x)
        return self._run(x)

    def _autograd_forward(self, x):
        out = self._run(x)
        return out, (self._plan_for(x), out, bool(x.requires_grad))

    def _autograd_backward(self, state, dout):
        """torch.autograd through dehaze22.py:205-362 / :364-488 (train mode; eval mode -- running statistics as constants, no dropout --
        since round 6): the plan walked in reverse -- every transposed conv as
        its four parity convolutions (their filter gradients gathered back into the (cin, cout, 4, 4) parameter), the side-by-side
        BatchNorm tables' dgamma / dbeta handed to the two modules each table was built from, BatchNorm + Dropout2d and the pooling
        head on csrc/legacy_bwd.hip."""
        P, out, need_dx = state
        B = _plan_backward(P)
        B.zero_()
        grads = {}
        self._seed_head(P, B, out, dout)
        B.run(grads, skip_dx_of={P.xin.data_ptr()})
        self._derived_grads(P, grads)
        self._table_grads(P, grads)
        dx = None
        if need_dx:      # round 6: the gradient w.r.t. the input image, as models.dehaze22.D forms it -- layer1 reads the raw image (no
            # prologue), so it is the 4x4 stride-2 conv's data gradient of layer1's output gradient, on the any-stride direct kernel
            n, h, w, _ = P.xin.shape          # NHWC
            y1 = E.View(P.cat[1], self.cd[2], self.ce[0])
            dx = torch.empty((n, self.input_nc, h, w), dtype=torch.float32, device=out.device)
            E.conv_bwd_data_direct(B.G(y1).fd, self.layer1.layer1.weight.detach().contiguous(), E.conv_desc(4, 2, 1, cout=self.ce[0]), dx)
        return dx, grads


class G(_UNet):
    """dehaze22.py:205-362: U-Net -> 20 channels -> four-scale pooling head (avg_pool 16 / 8 / 4 / 2, Conv2d(20, 1, 1),
    LeakyReLU, nearest upsampling, :343-354) -> Conv2d(24, output_nc, 3, 1, 1) -> tanh.  (B,3,H,W) -> (B,output_nc,H,W)."""

    def __init__(self, input_nc, output_nc, nf):
        super().__init__(input_nc, output_nc, nf)
        for nm in ("conv1010", "conv1020", "conv1030", "conv1040"):
            setattr(self, nm, nn.Conv2d(20, 1, kernel_size=1, stride=1, padding=0))
        self.refine3 = nn.Conv2d(20 + 4, 3, kernel_size=3, stride=1, padding=1)     # registered, never called (:316)
        self._make_layers(20)
        self.dlayerfinal = _Named(dlayer1=_Named(conv=nn.Conv2d(24, output_nc, 3, 1, 1, bias=False), tanh=nn.Tanh()))
        self.relu = nn.LeakyReLU(0.2, inplace=True)

    def _build_head(self, P, n, h, w, dev):
        nf = self.nf
        head = E.new_act(n, h, w, 24, dev)          # [dout1 (20) | x1010 x1020 x1030 x1040]: the reference's order is pyramid first,
        P.head = head                                # the final filter's input channels are permuted instead (16-byte alignment)
        self._tconv(P, E.View(P.cat[1]), P.last_pro, self.dlayer1.dlayer1.tconv.weight, 20, 2 * nf, head, 0, h // 2, w // 2,
                    L.ACT_NONE, None, 0, "dlayer1")
        P.pw = torch.zeros((4, 20), dtype=torch.float32, device=dev)
        P.pb = torch.zeros((4,), dtype=torch.float32, device=dev)
        for i, nm in enumerate(("conv1010", "conv1020", "conv1030", "conv1040")):
            conv = getattr(self, nm)
            P.copies += [(P.pw[i], conv.weight.view(20)), (P.pb[i:i + 1], conv.bias)]
        x20, y4 = E.View(head, 0, 20), E.View(head, 20, 4)
        P.op(lambda: E.pyramid_pool4(x20, P.pw, P.pb, 16, 0.2, y4),
             record=dict(kind="pyramid", src=x20, dst=y4, w=P.pw, b=P.pb, k0=16, slope=0.2,
                         sink=_pyramid_sink([getattr(self, nm) for nm in ("conv1010", "conv1020", "conv1030", "conv1040")])))
        P.wfinal = torch.zeros((self.output_nc, 24, 3, 3), dtype=torch.float32, device=dev)
        P.wfinal.requires_grad_(self.dlayerfinal.dlayer1.conv.weight.requires_grad)
        P.out = torch.empty((n, self.output_nc, h, w), dtype=torch.float32, device=dev)
        P.conv(E.View(head), P.weight(P.wfinal, self.output_nc, 24, 3), None, 3, pad=1, e_act=L.ACT_TANH, y_fd=E.nchw_f32_view(P.out),
               label="dlayerfinal")
        P.keep += [x20, y4]

    def _run(self, x):
        P = self._plan_for(x)
        wf = self.dlayerfinal.dlayer1.conv.weight
        P.wfinal.requires_grad_(wf.requires_grad)
        with torch.no_grad():                        # reference channel order [pyramid 0-3 | dout1 4-23] -> buffer order
            if getattr(P.wfinal, "_src_version", None) != (wf._version, wf.data_ptr()):
                P.wfinal[:, :20].copy_(wf.detach()[:, 4:])
                P.wfinal[:, 20:].copy_(wf.detach()[:, :4])
                P.wfinal._src_version = (wf._version, wf.data_ptr())
        return super()._run(x)

    def _finish(self, P, x):
        return P.out.clone()

    def _seed_head(self, P, B, out, dout):
        n, c, h, w = out.shape
        g8 = E.new_grad(n, h, w, (c + 7) // 8 * 8, out.device)
        E.out_act_bwd(dout, out, L.ACT_TANH, E.View(g8))          # tanh(dlayerfinal(.)) (:356-361)
        last = P.records[-1]
        assert last["kind"] == "conv" and last["y"] is None
        last["_dy"] = E.View(g8, 0, c)

    def _derived_grads(self, P, grads):
        super()._derived_grads(P, grads)
        _permuted_final_grad(P, grads, self.dlayerfinal.dlayer1.conv)


class G2(_UNet):
    """dehaze22.py:364-488: the same U-Net ending in ConvTranspose2d(2 nf, output_nc) + LeakyReLU(0.2) (:384-386)."""

    def __init__(self, input_nc, output_nc, nf):
        super().__init__(input_nc, output_nc, nf)
        self._make_layers(output_nc)
        self.dlayer1.dlayer1.add_module("tanh", nn.LeakyReLU(0.2, inplace=True))     # named `tanh` in the reference too

    def _build_head(self, P, n, h, w, dev):
        r8 = (self.output_nc + 7) // 8 * 8
        P.head = E.new_act(n, h, w, r8, dev, zero=True)
        self._tconv(P, E.View(P.cat[1]), P.last_pro, self.dlayer1.dlayer1.tconv.weight, self.output_nc, 2 * self.nf, P.head, 0, h // 2, w // 2,
                    L.ACT_LEAKY02, None, 0, "dlayer1")

    def _finish(self, P, x):
        out = torch.empty((x.shape[0], self.output_nc, x.shape[2], x.shape[3]), dtype=torch.float32, device=x.device)
        E.to_nchw(E.View(P.head, 0, self.output_nc), out)
        return out

    def _seed_head(self, P, B, out, dout):
        """The output IS the head buffer (LeakyReLU epilogue of dlayer1's parity convolutions): its gradient is dout, NHWC bf16;
        the epilogue's mask is applied by the walk."""
        E.to_nhwc(dout, E.View(B.gbuf[P.head.data_ptr()]))          # the padding channels of the 8-channel buffer become zeros


class Dense(_DensePyramid):
    """dehaze22.py:531-660: the DCPDN transmission network (the same network as dehaze1113.Dense2)."""


class dehaze(_PlannedModule):
    """dehaze22.py:662-753 (forward, and backward in train mode): transmission t = Dense(x), airlight A = G2(x) pooled over H x H windows, the
    scattering model inverted per pixel, J = (x - A) / (|t| + 1e-10) + A, then refine1 / refine2, the four-scale head and
    tanh(refine3).  Returns (dehaze, tran, atp, dehaze2) like the reference.  `tran_est` (a G) is registered and never called
    (:665), as there.  The two sub-networks run their own plans; the plan here is the tail."""

    def __init__(self, input_nc, output_nc, nf):
        super().__init__()
        self.tran_est = G(input_nc=3, output_nc=3, nf=64)
        self.atp_est = G2(input_nc=3, output_nc=3, nf=8)
        self.tran_dense = Dense()
        self.relu = nn.LeakyReLU(0.2, inplace=True)
        self.tanh = nn.Tanh()
        self.refine1 = nn.Conv2d(6, 20, kernel_size=3, stride=1, padding=1)
        self.refine2 = nn.Conv2d(20, 20, kernel_size=3, stride=1, padding=1)
        self.threshold = nn.Threshold(0.1, 0.1)
        for nm in ("conv1010", "conv1020", "conv1030", "conv1040"):
            setattr(self, nm, nn.Conv2d(20, 1, kernel_size=1, stride=1, padding=0))
        self.refine3 = nn.Conv2d(20 + 4, 3, kernel_size=3, stride=1, padding=1)
        self.upsample = nn.functional.interpolate
        self.batch1 = nn.BatchNorm2d(20)               # registered, never called (:686)

    def _build_plan(self, shape, dev):
        n, c, h, w = shape
        P = NetPlan(dev)
        P.cat6 = E.new_act(n, h, w, 8, dev, zero=True)            # [J (3) | x (3) | 0 0]
        P.r1 = E.new_act(n, h, w, 24, dev, zero=True)
        P.head = E.new_act(n, h, w, 24, dev)                      # [refine2 output (20) | pyramid (4)]
        P.conv(E.View(P.cat6, 0, 6), P.weight(self.refine1.weight, 20, 6, 3), E.View(P.r1, 0, 20), 3, pad=1, bias=self.refine1.bias,
               e_act=L.ACT_LEAKY02, label="refine1")
        P.conv(E.View(P.r1, 0, 20), P.weight(self.refine2.weight, 20, 20, 3), E.View(P.head, 0, 20), 3, pad=1, bias=self.refine2.bias,
               e_act=L.ACT_LEAKY02, label="refine2")
        P.pw = torch.zeros((4, 20), dtype=torch.float32, device=dev)
        P.pb = torch.zeros((4,), dtype=torch.float32, device=dev)
        x20, y4 = E.View(P.head, 0, 20), E.View(P.head, 20, 4)
        P.op(lambda: E.pyramid_pool4(x20, P.pw, P.pb, 32, 0.2, y4),
             record=dict(kind="pyramid", src=x20, dst=y4, w=P.pw, b=P.pb, k0=32, slope=0.2,
                         sink=_pyramid_sink([getattr(self, nm) for nm in ("conv1010", "conv1020", "conv1030", "conv1040")])))
        P.wfinal = torch.zeros((3, 24, 3, 3), dtype=torch.float32, device=dev)
        P.wfinal.requires_grad_(self.refine3.weight.requires_grad)
        P.out = torch.empty((n, 3, h, w), dtype=torch.float32, device=dev)
        P.conv(E.View(P.head), P.weight(P.wfinal, 3, 24, 3), None, 3, pad=1, bias=self.refine3.bias, e_act=L.ACT_TANH,
               y_fd=E.nchw_f32_view(P.out), label="refine3")
        P.wmean = torch.zeros((n * 3 * max(w // h, 1),), dtype=torch.float32, device=dev)
        P.keep += [x20, y4]
        return P.finish()

    def forward(self, x):
        if x.shape[3] < x.shape[2]:
            raise ValueError("dehaze pools the airlight over H x H windows (dehaze22.py:705): W >= H required, got %dx%d" % tuple(x.shape[2:]))
        if _wants_grad(self, x):
            # autograd composes the three pieces: the two sub-networks are planned modules with their own reverse walks, the tail
            # (scattering model + refinement) is one more autograd.Function; `tran` is returned as the sub-network produced it
            # an input image that requires grad (round 6) stays in the graph: both sub-networks return its gradient through their
            # own reverse walks, the tail returns the scattering model's and the refinement's (`_tail_backward`), autograd adds them
            xf = x.float().contiguous() if x.requires_grad else x.detach().float().contiguous()
            tran = self.tran_dense(xf)
            atp_raw = self.atp_est(xf)
            params = tuple(p for nm, p in self.named_parameters() if p.requires_grad and not nm.startswith(("tran_dense.", "atp_est.", "tran_est.")))
            out, atp, dehaze2 = _DehazeTail.apply(self, xf, tran, atp_raw, *params)
            return out, tran, atp, dehaze2
        with torch.no_grad():
            xf = x.detach().float().contiguous()
            tran = self.tran_dense(xf)
            atp_raw = self.atp_est(xf)
            return self._tail(xf, tran, atp_raw)[1]

    def _tail(self, xf, tran, atp_raw):
        """Scattering model + refinement on finished sub-network outputs (no autograd): (plan, (dehaze, tran, atp, dehaze2))."""
        with torch.no_grad():
            P = self._plan_for(xf)
            for i, nm in enumerate(("conv1010", "conv1020", "conv1030", "conv1040")):
                conv = getattr(self, nm)
                P.pw[i].copy_(conv.weight.detach().view(20))
                P.pb[i:i + 1].copy_(conv.bias.detach())
            wf = self.refine3.weight
            P.wfinal.requires_grad_(wf.requires_grad)
            if getattr(P.wfinal, "_src_version", None) != (wf._version, wf.data_ptr()):
                P.wfinal[:, :20].copy_(wf.detach()[:, 4:])       # reference order: [pyramid 0-3 | features 4-23]
                P.wfinal[:, 20:].copy_(wf.detach()[:, :4])
                P.wfinal._src_version = (wf._version, wf.data_ptr())
            atp, dehaze2 = torch.empty_like(xf), torch.empty_like(xf)
            E.scatter_dehaze(xf, tran, atp_raw, 0.2, 1e-10, P.wmean, atp, dehaze2, E.View(P.cat6))
            P.launch()
            return P, (P.out.clone(), tran, atp, dehaze2)

    def _tail_backward(self, P, xf, tran, atp_raw, out, g_out, g_atp, g_dehaze2, need_dx=False):
        """Reverse of `_tail`: the refinement plan walked in reverse, then the scattering model (csrc/legacy_bwd.hip).  need_dx: also the
        tail's own gradient w.r.t. the image -- dJ/dI = 1 / (|t| + eps) on J's total gradient, plus the image's copy in the
        refinement's input [J | x] (three-channel elementwise glue on the gradient buffer the walk left)."""
        B = _plan_backward(P)
        B.zero_()
        n, c, h, w = out.shape
        g8 = E.new_grad(n, h, w, 8, out.device)
        E.out_act_bwd(g_out, out, L.ACT_TANH, E.View(g8))                  # tanh(refine3(.)) (:752)
        last = P.records[-1]
        assert last["kind"] == "conv" and last["y"] is None
        last["_dy"] = E.View(g8, 0, 3)
        grads = {}
        B.run(grads)
        _permuted_final_grad(P, grads, self.refine3)
        d_tran, d_atp = E.scatter_dehaze_bwd(xf, tran, atp_raw, P.wmean, 0.2, 1e-10, g_dehaze2, g_atp, B.G(E.View(P.cat6)))
        if not need_dx:
            return d_tran, d_atp, grads
        gcat = B.G(E.View(P.cat6, 0, 6)).torch_nchw()
        d_x = (gcat[:, :3] + g_dehaze2) / (tran.abs() + 1e-10) + gcat[:, 3:]
        return d_tran, d_atp, grads, d_x.contiguous()


class _DehazeTail(torch.autograd.Function):
    """The part of `dehaze` between its sub-networks' outputs and its own: (x, tran, atp_raw) -> (dehaze, atp, dehaze2)."""

    @staticmethod
    def forward(ctx, module, xf, tran, atp_raw, *params):
        tran_c, atp_c = tran.detach().contiguous(), atp_raw.detach().contiguous()
        ctx.need_dx = bool(xf.requires_grad)
        xf = xf.detach()
        P, (out, _, atp, dehaze2) = module._tail(xf, tran_c, atp_c)
        ctx.module, ctx.plan, ctx.params = module, P, params
        ctx.gen = _bump_generation(P)
        ctx.save_for_backward(xf, tran_c, atp_c, out)
        return out, atp, dehaze2

    @staticmethod
    def backward(ctx, g_out, g_atp, g_dehaze2):
        _check_generation(ctx.plan, ctx.gen)
        xf, tran, atp_raw, out = ctx.saved_tensors
        f = lambda g: g.detach().float().contiguous()
        res = ctx.module._tail_backward(ctx.plan, xf, tran, atp_raw, out, f(g_out), f(g_atp), f(g_dehaze2), ctx.need_dx)
        d_tran, d_atp, grads = res[:3]
        return (None, res[3] if ctx.need_dx else None, d_tran, d_atp) + autograd_grads(grads, ctx.params)
