"""TEST INFRASTRUCTURE (oracle).  Generates tests/golden/*.npz + MANIFEST.json.

Run in the BUILD container only (needs /root/reference):
    python -m oracle.make_golden

For every component it (1) runs the real reference module (imported behind the
shims of oracle/ref_import.py) and the oracle restatement on the same seeded
input and name-keyed deterministic weights, (2) records max|ref - oracle| in the
manifest (must be ~0: same torch CPU ops), and (3) stores the REFERENCE's outputs
as the golden vectors.  Blur/Laplacian have no importable reference (bytecode
only): their vectors come from the restatement and are pinned by the
known-answer values listed in the manifest.

Fixtures are data only: inputs are regenerated from seeds, outputs are stored.
"""
import json
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, "tests", "golden")


def _sub(t, c=4, s=4):
    """Channel/space strided subsample so activation taps stay small."""
    return t[:, ::c, ::s, ::s].contiguous().numpy()


def _copy_weights(dst, src):
    with torch.no_grad():
        for (k, v), (k2, v2) in zip(dst.state_dict().items(), src.state_dict().items()):
            assert k == k2, (k, k2)
            v.copy_(v2)


def main():
    warnings.filterwarnings("ignore")
    sys.path.insert(0, ROOT)
    from oracle.ref_import import import_reference, import_reference_metrics
    from oracle import dehaze1113_ref as o1113
    from oracle.vgg16_ref import Vgg16 as OVgg
    from oracle import ssim_ref, freqsplit_ref
    from oracle.detweights import fill_state_dict, det_input
    r1113, r22, RVgg, rssim = import_reference()
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    man = {"torch": torch.__version__, "numpy": np.__version__, "ref_vs_oracle_maxabs": {}, "kat": {}}

    # ---------------- FDGAN forward + backward, 2x3x64x64 ----------------
    og, rg = o1113.FDGAN(), r1113.FDGAN()
    fill_state_dict(og, seed=0)
    _copy_weights(rg, og)
    x = det_input((2, 3, 64, 64), seed=1234)
    tgt = det_input((2, 3, 64, 64), seed=4321, lo=-1.0, hi=1.0)
    taps = {}
    xo = x.clone().requires_grad_(True)
    yo = og(xo, taps)
    (((yo - tgt) ** 2).mean()).backward()
    xr = x.clone().requires_grad_(True)
    yr = rg(xr)
    (((yr - tgt) ** 2).mean()).backward()
    man["ref_vs_oracle_maxabs"]["fdgan_fwd"] = float((yo - yr).abs().max())
    man["ref_vs_oracle_maxabs"]["fdgan_dx"] = float((xo.grad - xr.grad).abs().max())
    gnames = ["conv_refin1.weight", "conv_refin1.bias", "conv_refin3.weight", "conv_refin3.bias",
              "conv_refine4.bias", "dense_block1.denselayer1.norm1.weight",
              "dense_block1.denselayer1.norm1.bias", "dense_block1.denselayer6.conv2.weight",
              "dense_block3.denselayer24.conv1.weight", "trans_block2.norm.weight",
              "trans_block4.conv1.weight", "dense_block6.conv2.weight"]
    rp, op = dict(rg.named_parameters()), dict(og.named_parameters())
    gd = max(float((rp[n].grad - op[n].grad).abs().max()) for n in gnames)
    man["ref_vs_oracle_maxabs"]["fdgan_dparams"] = gd
    unused = sorted(n for n, p in rp.items() if p.grad is None)
    man["fdgan_params_without_grad"] = len(unused)
    man["fdgan_numel_with_grad"] = int(sum(p.numel() for p in rp.values() if p.grad is not None))
    # post-forward BN buffers (train-mode side effects)
    rb = dict(rg.named_buffers())
    bn = {"bn_rm__" + n.replace(".", "__"): rb[n + ".running_mean"].numpy()
          for n in ["dense_block1.denselayer1.norm1", "dense_block2.denselayer12.norm2", "trans_block3.norm"]}
    bn.update({"bn_rv__" + n.replace(".", "__"): rb[n + ".running_var"].numpy()
               for n in ["dense_block1.denselayer1.norm1", "dense_block2.denselayer12.norm2", "trans_block3.norm"]})
    bn["nbt"] = rb["dense_block1.denselayer1.norm1.num_batches_tracked"].numpy()
    np.savez_compressed(
        os.path.join(OUT, "fdgan_2x64.npz"), y=yr.detach().numpy(), dx=xr.grad.numpy(),
        **{"tap__" + k: _sub(v) for k, v in taps.items()},
        **{"tapstat__" + k: np.array([v.mean().item(), v.std().item()]) for k, v in taps.items()},
        **{"grad__" + n.replace(".", "__"): rp[n].grad.numpy() for n in gnames}, **bn)

    # ---------------- FDGAN backward, well-conditioned: batch 8 @ 64x64, BatchNorm biases + 3 ----------------
    # (oracle/detweights.shift_bn_bias: ReLU mask flips under bf16 rounding become rare, so ALL 282 parameter
    # gradients are comparable.)  Stored: per parameter 64 signed strided sums + the norm of the REFERENCE's gradient,
    # and the oracle-vs-bf16-emulating-oracle disagreement of that parameter (the noise floor a bf16 path can reach).
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from hiputil import emulate_kernel_operands, rel_rms
    from oracle.detweights import grad_projection, shift_bn_bias
    ogw, rgw, oew = o1113.FDGAN(), r1113.FDGAN(), o1113.FDGAN()
    fill_state_dict(ogw, seed=0)
    shift_bn_bias(ogw, 3.0)
    _copy_weights(rgw, ogw)
    oew.load_state_dict(ogw.state_dict())
    emulate_kernel_operands(oew, round_grads=True)      # fp16 forward operands AND bf16-stored activation gradients: what the HIP path does
    xw = det_input((8, 3, 64, 64), seed=1234)
    tw = det_input((8, 3, 64, 64), seed=4321, lo=-1.0, hi=1.0)
    yrw = rgw(xw.clone())
    ((yrw - tw) ** 2).mean().backward()
    ((ogw(xw.clone()) - tw) ** 2).mean().backward()
    ((oew(xw.clone()) - tw) ** 2).mean().backward()
    rpw, opw, epw = dict(rgw.named_parameters()), dict(ogw.named_parameters()), dict(oew.named_parameters())
    wc, o2o = {}, {}
    for n, p in rpw.items():
        if p.grad is None:
            continue
        proj, norm = grad_projection(n, p.grad.numpy())
        key = n.replace(".", "__")
        o2o[n] = rel_rms(epw[n].grad, opw[n].grad)
        wc["proj__" + key], wc["norm__" + key], wc["o2o__" + key] = proj, np.float64(norm), np.float64(o2o[n])
    man["ref_vs_oracle_maxabs"]["fdgan_wellcond_dparams"] = max(float((rpw[n].grad - opw[n].grad).abs().max()) for n in o2o)
    man["ref_vs_oracle_maxabs"]["fdgan_wellcond_fwd"] = float((yrw - ogw(xw.clone())).abs().max())
    vals_o2o = np.array(list(o2o.values()))
    man["fdgan_wellcond"] = {"params": len(o2o), "bn_bias_shift": 3.0, "batch": 8, "size": 64,
                             "oracle_vs_emulated_median": float(np.median(vals_o2o)),
                             "oracle_vs_emulated_p90": float(np.percentile(vals_o2o, 90)),
                             "oracle_vs_emulated_worst": sorted(((float(v), k) for k, v in o2o.items()), reverse=True)[:5]}
    np.savez_compressed(os.path.join(OUT, "fdgan_8x64_wellcond.npz"), y=yrw.detach().numpy()[:, :, ::4, ::4], **wc)

    # eval-mode forward (running statistics) on the same weights, fresh modules
    og2, rg2 = o1113.FDGAN().eval(), r1113.FDGAN().eval()
    fill_state_dict(og2, seed=0)
    _copy_weights(rg2, og2)
    with torch.no_grad():
        ye_o, ye_r = og2(x.clone()), rg2(x.clone())
    man["ref_vs_oracle_maxabs"]["fdgan_fwd_eval"] = float((ye_o - ye_r).abs().max())
    np.savez_compressed(os.path.join(OUT, "fdgan_2x64_eval.npz"), y=ye_r.numpy())

    # ---------------- decoder blocks standalone ----------------
    ob, rb_ = o1113.BottleneckBlockdy(64, 32), r1113.BottleneckBlockdy(64, 32)
    fill_state_dict(ob, seed=3)
    _copy_weights(rb_, ob)
    xb = det_input((2, 64, 16, 16), seed=5, lo=-1.0, hi=1.0)
    xb_o, xb_r = xb.clone(), xb.clone()
    with torch.no_grad():
        yb_o, yb_r = ob(xb_o), rb_(xb_r)
    man["ref_vs_oracle_maxabs"]["bottleneckdy"] = float((yb_o - yb_r).abs().max())
    man["ref_vs_oracle_maxabs"]["bottleneckdy_inplace_x"] = float((xb_o - xb_r).abs().max())
    ot, rt = o1113.TransitionBlockdy(96, 16), r1113.TransitionBlockdy(96, 16)
    fill_state_dict(ot, seed=4)
    _copy_weights(rt, ot)
    with torch.no_grad():
        yt_o, yt_r = ot(yb_o.clone()), rt(yb_r.clone())
    man["ref_vs_oracle_maxabs"]["transitiondy"] = float((yt_o - yt_r).abs().max())
    np.savez_compressed(os.path.join(OUT, "dyblocks.npz"), y_bottleneck=yb_r.numpy(),
                        x_after=xb_r.numpy(), y_transition=yt_r.numpy())
    # the same two blocks with dropRate > 0 in training mode (/root/reference/models/dehaze1113.py:270-274, :367-368; FDGAN itself
    # passes 0): the REAL modules under a seed give the outputs, the oracle under the same seed must give the same bits and hands
    # out the masks F.dropout drew (stored as bits), so that the HIP path can be run on exactly those masks
    obd, rbd = o1113.BottleneckBlockdy(64, 32, 0.3), r1113.BottleneckBlockdy(64, 32, 0.3)
    fill_state_dict(obd, seed=3)
    _copy_weights(rbd, obd)
    otd, rtd = o1113.TransitionBlockdy(96, 16, 0.25), r1113.TransitionBlockdy(96, 16, 0.25)
    fill_state_dict(otd, seed=4)
    _copy_weights(rtd, otd)
    with torch.no_grad():
        torch.manual_seed(2024)
        ybd_r = rbd(xb.clone())
        ytd_r = rtd(ybd_r.clone())
        torch.manual_seed(2024)
        ybd_o = obd(xb.clone())
        ytd_o = otd(ybd_o.clone())
    man["ref_vs_oracle_maxabs"]["bottleneckdy_dropout_train"] = float((ybd_o - ybd_r).abs().max())
    man["ref_vs_oracle_maxabs"]["transitiondy_dropout_train"] = float((ytd_o - ytd_r).abs().max())
    np.savez_compressed(os.path.join(OUT, "dyblocks_dropout.npz"), y_bottleneck=ybd_r.numpy(), y_transition=ytd_r.numpy(),
                        mask_b0=np.packbits(obd.masks[0].numpy().astype(np.uint8)), mask_b1=np.packbits(obd.masks[1].numpy().astype(np.uint8)),
                        mask_t0=np.packbits(otd.masks[0].numpy().astype(np.uint8)),
                        shape_b0=np.array(obd.masks[0].shape), shape_b1=np.array(obd.masks[1].shape), shape_t0=np.array(otd.masks[0].shape))

    # ---------------- Fusion-D (9,36) and dehaze22.D, 2x9x64x64 ----------------
    od, rd = o1113.D(9, 36), r1113.D(9, 36)
    fill_state_dict(od, seed=1)
    _copy_weights(rd, od)
    xd = det_input((2, 9, 64, 64), seed=77, lo=-1.0, hi=1.0)
    xdo = xd.clone().requires_grad_(True)
    ydo = od(xdo)
    ydo.mean().backward()
    xdr = xd.clone().requires_grad_(True)
    ydr = rd(xdr)
    ydr.mean().backward()
    man["ref_vs_oracle_maxabs"]["d_fwd"] = float((ydo - ydr).abs().max())
    man["ref_vs_oracle_maxabs"]["d_dx"] = float((xdo.grad - xdr.grad).abs().max())
    rdp = {k: v for k, v in zip([k for k, _ in od.named_parameters()], [p for p in rd.parameters()])}
    np.savez_compressed(os.path.join(OUT, "d_2x64.npz"), y=ydr.detach().numpy(), dx=xdr.grad.numpy(),
                        **{"grad__" + k.replace(".", "__"): v.grad.numpy() for k, v in rdp.items()
                           if v.numel() < 30000})
    rd22 = r22.D(9, 36)
    sd22 = fill_state_dict(rd22, seed=2)       # dotted names: fill works on state_dict directly
    with torch.no_grad():
        y22 = rd22(xd.clone())
    np.savez_compressed(os.path.join(OUT, "d22_2x64.npz"), y=y22.numpy())
    man["d22_keys"] = list(sd22.keys())

    # ---------------- legacy U-Nets dehaze22.G / G2, 2x3x256x256, nf = 8 ----------------
    from . import legacy_ref
    xl = det_input((2, 3, 256, 256), seed=21)
    for kind, cls in (("G", r22.G), ("G2", r22.G2)):
        net = cls(3, 3, 8)
        fill_state_dict(net, seed=5)
        for kk, vv in net.state_dict().items():         # the generic fill saturates G's tanh: keep the last filter small
            if kk == "dlayerfinal.dlayer1.conv.weight":
                vv.mul_(0.3)
        sd0 = {kk: vv.clone() for kk, vv in net.state_dict().items()}
        net.eval()
        with torch.no_grad():
            ye = net(xl.clone())
            yo, _ = legacy_ref.unet_forward({kk: vv.clone() for kk, vv in sd0.items()}, xl.clone(), False, kind)
        man["ref_vs_oracle_maxabs"]["legacy_%s_eval" % kind] = float((ye - yo).abs().max())
        net.train()
        sdt = {kk: vv.clone() for kk, vv in sd0.items()}
        with torch.no_grad():
            torch.manual_seed(77)
            yt = net(xl.clone())
            torch.manual_seed(77)
            yot, masks = legacy_ref.unet_forward(sdt, xl.clone(), True, kind)
        man["ref_vs_oracle_maxabs"]["legacy_%s_train" % kind] = float((yt - yot).abs().max())
        after = net.state_dict()
        np.savez_compressed(os.path.join(OUT, "legacy_%s_2x256.npz" % kind.lower()), y_eval=ye.numpy()[:, :, ::4, ::4],
                            y_train=yt.numpy()[:, :, ::4, ::4], masks=torch.stack(masks).numpy(),
                            rm_dlayer5=after["dlayer5.dlayer5.bn.running_mean"].numpy(),
                            rv_dlayer5=after["dlayer5.dlayer5.bn.running_var"].numpy(),
                            rm_layer8=after["layer8.layer8.bn.running_mean"].numpy(),
                            rv_layer8=after["layer8.layer8.bn.running_var"].numpy())
        man["legacy_%s_keys" % kind] = list(sd0.keys())

    # ---------------- legacy `Dense` (three spellings of one topology), 2x3x128x160 ----------------
    xd_ = det_input((2, 3, 128, 160), seed=31)
    for nm, cls, tail in (("dense1113", r1113.Dense, "bn"), ("dense2_1113", r1113.Dense2, "pyramid"), ("dense22", r22.Dense, "pyramid")):
        net = cls()
        fill_state_dict(net, seed=6)
        with torch.no_grad():                        # keep the tanh unsaturated
            net.refine3.weight.mul_(0.1), net.refine3.bias.mul_(0.1)
        sd0 = {kk: vv.clone() for kk, vv in net.state_dict().items()}
        outs = {}
        for mode in (False, True):
            net.load_state_dict(sd0)
            net.train(mode)
            sdm = {kk: vv.clone() for kk, vv in sd0.items()}
            with torch.no_grad():
                yr = net(xd_.clone())
                yo = legacy_ref.dense_forward(sdm, xd_.clone(), mode, tail)
            man["ref_vs_oracle_maxabs"]["legacy_%s_%s" % (nm, "train" if mode else "eval")] = float((yr - yo).abs().max())
            outs["y_train" if mode else "y_eval"] = yr.numpy()[:, :, ::2, ::2]
            if mode:
                after = net.state_dict()
                outs["rm_norm0"], outs["rv_norm0"] = after["norm0.running_mean"].numpy(), after["norm0.running_var"].numpy()
                outs["rm_tb5"], outs["rv_tb5"] = after["trans_block5.bn1.running_mean"].numpy(), after["trans_block5.bn1.running_var"].numpy()
        np.savez_compressed(os.path.join(OUT, "legacy_%s_2x128.npz" % nm), **outs)
        man["legacy_%s_keys" % nm] = len(sd0)

    # ---------------- legacy `dehaze` (dehaze22.py:662-753): Dense + G2 + the scattering model, 2x3x256x256 ----------------
    net = r22.dehaze(3, 3, 64)
    fill_state_dict(net, seed=8)
    with torch.no_grad():       # transmission bounded away from zero (J divides by |t|), final tanh unsaturated
        net.tran_dense.refine3.weight.mul_(0.05), net.tran_dense.refine3.bias.fill_(1.0), net.refine3.weight.mul_(0.02)
    sd0 = {kk: vv.clone() for kk, vv in net.state_dict().items()}
    xh = det_input((2, 3, 256, 256), seed=41)
    outs = {}
    for mode in (False, True):
        with torch.no_grad():
            for kk, vv in net.state_dict().items():     # load_state_dict cannot resolve the dotted child names
                vv.copy_(sd0[kk])
        net.train(mode)
        sdm = {kk: vv.clone() for kk, vv in sd0.items()}
        with torch.no_grad():
            torch.manual_seed(3)
            yr = net(xh.clone())
            torch.manual_seed(3)
            yo = legacy_ref.dehaze_forward(sdm, xh.clone(), mode)
        tag = "train" if mode else "eval"
        man["ref_vs_oracle_maxabs"]["legacy_dehaze_" + tag] = max(float((a - b).abs().max()) for a, b in zip(yr, yo[:4]))
        for nm_, t_ in zip(("dehaze", "tran", "atp", "dehaze2"), yr):
            outs[nm_ + "_" + tag] = t_.numpy()[:, :, ::8, ::8]
        if mode:
            outs["masks"] = torch.stack(yo[4]).numpy()
    np.savez_compressed(os.path.join(OUT, "legacy_dehaze_2x256.npz"), **outs)
    man["legacy_dehaze_keys"] = len(sd0)

    # ---------------- VGG16 features, 1x3x32x32 ----------------
    ov, rv = OVgg(), RVgg()
    fill_state_dict(ov, seed=0)
    rv.load_state_dict(ov.state_dict())
    xv = det_input((1, 3, 32, 32), seed=9)
    with torch.no_grad():
        fo, fr = ov(xv), rv(xv)
    man["ref_vs_oracle_maxabs"]["vgg16"] = max(float((a - b).abs().max()) for a, b in zip(fo, fr))
    np.savez_compressed(os.path.join(OUT, "vgg16_1x32.npz"),
                        **{"relu%d" % i: f.numpy() for i, f in enumerate(fr)})

    # ---------------- SSIM loss ----------------
    a1, a2 = det_input((2, 3, 64, 64), seed=1), det_input((2, 3, 64, 64), seed=2)
    a3 = (a1 * 0.9 + 0.1 * a2)
    vals = {}
    for nm, (p, q) in {"rand": (a1, a2), "near": (a1, a3), "same": (a1, a1)}.items():
        vo, vr = ssim_ref.ssim(p, q).item(), rssim.ssim(p, q).item()
        vm = rssim.SSIM()(p, q).item()
        man["ref_vs_oracle_maxabs"]["ssim_" + nm] = abs(vo - vr)
        vals[nm] = [vr, vm]
    man["kat"]["pytorch_ssim"] = vals

    # ---------------- frequency split (no importable reference) ----------------
    xf = det_input((2, 3, 40, 48), seed=11)
    k = freqsplit_ref.isotropic_gaussian_kernel(15, 3.0)
    g1 = k.sum(1)
    man["kat"]["gauss15_sigma3"] = {"sum": float(k.sum()), "centre": float(k[7, 7]), "corner": float(k[0, 0]),
                                    "separable_maxabs": float(np.abs(k - np.outer(g1, g1)).max())}
    c = torch.full((1, 3, 20, 20), 0.7)
    man["kat"]["blur_const"] = float((freqsplit_ref.blur(c, use_input_norm=False) - 0.7).abs().max())
    lc = freqsplit_ref.laplacian(torch.ones(1, 3, 8, 8))
    man["kat"]["laplacian_ones"] = {"interior": float(lc[0, 0, 3, 3]), "corner": float(lc[0, 0, 0, 0]),
                                    "edge": float(lc[0, 0, 0, 3])}
    np.savez_compressed(os.path.join(OUT, "freqsplit.npz"),
                        blur_norm=freqsplit_ref.blur(xf, use_input_norm=True).numpy(),
                        blur_raw=freqsplit_ref.blur(xf, use_input_norm=False).numpy(),
                        lap=freqsplit_ref.laplacian(xf).numpy(), gauss1d=g1)

    # ---------------- PSNR / SSIM scorer (reference functions exec'd) ----------------
    cs, pm = import_reference_metrics()
    rng = np.random.default_rng(0)
    a = rng.integers(0, 256, (64, 80, 3), dtype=np.uint8)
    b = np.clip(a.astype(int) + rng.integers(-20, 21, a.shape), 0, 255).astype(np.uint8)
    ss = float(np.mean([cs(a[..., i], b[..., i], gaussian_weights=True, use_sample_covariance=False)
                        for i in range(3)]))
    A, B = a.astype(float) / 255.0, b.astype(float) / 255.0
    man["kat"]["psnrssim"] = {"ssim": ss, "psnr": float(pm(A[1:-1, 1:-1], B[1:-1, 1:-1]))}

    with open(os.path.join(OUT, "MANIFEST.json"), "w") as f:
        json.dump(man, f, indent=1, sort_keys=True)
    print(json.dumps(man["ref_vs_oracle_maxabs"], indent=1))
    print(json.dumps(man["kat"], indent=1))
    tot = sum(os.path.getsize(os.path.join(OUT, n)) for n in os.listdir(OUT))
    print("golden bytes:", tot)


if __name__ == "__main__":
    main()
