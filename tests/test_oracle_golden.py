"""CPU tests: the oracle restatement (oracle/) against the golden vectors that
oracle/make_golden.py produced from the REAL reference modules (imported from
/root/reference in the build container).  The reference itself is not needed
to run these tests."""
import os

import numpy as np
import pytest
import torch

from oracle import dehaze1113_ref as o1113
from oracle import freqsplit_ref, ssim_ref
from oracle.detweights import det_input, fill_state_dict
from oracle.vgg16_ref import Vgg16

TOL = dict(rtol=2e-4, atol=2e-5)      # torch CPU conv summation order may differ across hosts


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_manifest_says_oracle_equals_reference(manifest):
    for k, v in manifest["ref_vs_oracle_maxabs"].items():
        assert v <= 1e-6, (k, v)
    assert manifest["fdgan_numel_with_grad"] == 11803155
    kat = manifest["kat"]
    assert abs(kat["psnrssim"]["ssim"] - 0.9875573750475048) < 1e-12
    assert abs(kat["psnrssim"]["psnr"] - 26.8562966261433) < 1e-10


def test_fdgan_state_dict_surface():
    g = o1113.FDGAN()
    sd = g.state_dict()
    assert len(sd) == 786 and sum(p.numel() for p in g.parameters()) == 13980691
    assert tuple(sd["trans_block4.conv1.weight"].shape) == (768, 128, 1, 1)   # ConvT layout
    for k in ("conv0.weight", "dense_block31.denselayer16.conv2.weight", "dense_norm31.weight",
              "dense_block4.bn1.weight", "trans_block4.bn1.running_mean", "conv_refine4.bias",
              "dense_block1.denselayer1.norm1.num_batches_tracked"):
        assert k in sd
    assert g.training


def test_fdgan_forward_backward_matches_golden(golden_dir):
    gold = _load(golden_dir, "fdgan_2x64.npz")
    g = o1113.FDGAN()
    fill_state_dict(g, seed=0)
    x = det_input((2, 3, 64, 64), seed=1234).requires_grad_(True)
    tgt = det_input((2, 3, 64, 64), seed=4321, lo=-1.0, hi=1.0)
    taps = {}
    y = g(x, taps)
    np.testing.assert_allclose(y.detach().numpy(), gold["y"], **TOL)
    for k, v in taps.items():
        np.testing.assert_allclose(v[:, ::4, ::4, ::4].numpy(), gold["tap__" + k], rtol=1e-3, atol=1e-4)
    ((y - tgt) ** 2).mean().backward()
    np.testing.assert_allclose(x.grad.numpy(), gold["dx"], rtol=2e-3, atol=1e-7)
    params = dict(g.named_parameters())
    for key in gold.files:
        if key.startswith("grad__"):
            p = params[key[6:].replace("__", ".")]
            scale = np.abs(gold[key]).max() + 1e-12
            assert np.abs(p.grad.numpy() - gold[key]).max() <= 2e-3 * scale, key
    bufs = dict(g.named_buffers())
    np.testing.assert_allclose(bufs["trans_block3.norm.running_var"].numpy(),
                               gold["bn_rv__trans_block3__norm"], rtol=1e-4)
    assert int(bufs["dense_block1.denselayer1.norm1.num_batches_tracked"]) == int(gold["nbt"]) == 1
    assert params["conv0.weight"].grad is None and params["dense_block4.bn1.weight"].grad is None


def test_fdgan_eval_mode_matches_golden(golden_dir):
    gold = _load(golden_dir, "fdgan_2x64_eval.npz")
    g = o1113.FDGAN().eval()
    fill_state_dict(g, seed=0)
    with torch.no_grad():
        y = g(det_input((2, 3, 64, 64), seed=1234))
    np.testing.assert_allclose(y.numpy(), gold["y"], **TOL)


def test_dy_blocks_match_golden(golden_dir):
    gold = _load(golden_dir, "dyblocks.npz")
    b = o1113.BottleneckBlockdy(64, 32)
    fill_state_dict(b, seed=3)
    x = det_input((2, 64, 16, 16), seed=5, lo=-1.0, hi=1.0)
    with torch.no_grad():
        y = b(x)
    np.testing.assert_allclose(y.numpy(), gold["y_bottleneck"], **TOL)
    np.testing.assert_array_equal(x.numpy(), gold["x_after"])          # in-place ReLU aliasing
    assert (x >= 0).all()
    t = o1113.TransitionBlockdy(96, 16)
    fill_state_dict(t, seed=4)
    with torch.no_grad():
        z = t(y)
    np.testing.assert_allclose(z.numpy(), gold["y_transition"], **TOL)
    assert z.shape == (2, 16, 32, 32)


def _dropout_masks(gold):
    return [torch.from_numpy(np.unpackbits(gold["mask_" + k])[:int(np.prod(gold["shape_" + k]))].reshape(tuple(gold["shape_" + k])).astype(np.float32))
            for k in ("b0", "b1", "t0")]


def test_dy_blocks_with_dropout_match_golden(golden_dir, manifest):
    """dropRate > 0 (/root/reference/models/dehaze1113.py:270-274, :367-368): the REAL blocks under seed 2024 wrote the fixture; the
    oracle's spelled-out F.dropout under the same seed gives the same bits and draws exactly the stored masks."""
    gold = _load(golden_dir, "dyblocks_dropout.npz")
    assert manifest["ref_vs_oracle_maxabs"]["bottleneckdy_dropout_train"] == 0.0 and manifest["ref_vs_oracle_maxabs"]["transitiondy_dropout_train"] == 0.0
    b, t = o1113.BottleneckBlockdy(64, 32, 0.3), o1113.TransitionBlockdy(96, 16, 0.25)
    fill_state_dict(b, seed=3), fill_state_dict(t, seed=4)
    x = det_input((2, 64, 16, 16), seed=5, lo=-1.0, hi=1.0)
    with torch.no_grad():
        torch.manual_seed(2024)
        y = b(x)
        z = t(y.clone())
    np.testing.assert_allclose(y.numpy(), gold["y_bottleneck"], **TOL)
    np.testing.assert_allclose(z.numpy(), gold["y_transition"], **TOL)
    for got, want in zip(b.masks + t.masks, _dropout_masks(gold)):
        assert torch.equal(got, want)
    frac = float(b.masks[0].mean())
    assert 0.65 < frac < 0.75                                            # keep probability 1 - 0.3
    b0 = o1113.BottleneckBlockdy(64, 32)                                 # eval mode: F.dropout(training=False) is the identity
    b0.load_state_dict(b.state_dict())
    b.eval(), b0.eval()
    with torch.no_grad():
        assert torch.equal(b(x.clone()), b0(x.clone())) and not b.masks


def test_fusion_d_matches_golden(golden_dir):
    gold = _load(golden_dir, "d_2x64.npz")
    d = o1113.D(9, 36)
    fill_state_dict(d, seed=1)
    assert sum(p.numel() for p in d.parameters()) == 790416
    x = det_input((2, 9, 64, 64), seed=77, lo=-1.0, hi=1.0).requires_grad_(True)
    y = d(x)
    assert y.shape == (2, 1, 30, 30)
    np.testing.assert_allclose(y.detach().numpy(), gold["y"], **TOL)
    y.mean().backward()
    np.testing.assert_allclose(x.grad.numpy(), gold["dx"], rtol=2e-3, atol=1e-9)


def test_vgg16_matches_golden(golden_dir):
    gold = _load(golden_dir, "vgg16_1x32.npz")
    v = Vgg16()
    fill_state_dict(v, seed=0)
    assert sum(p.numel() for p in v.parameters()) == 14714688
    with torch.no_grad():
        feats = v(det_input((1, 3, 32, 32), seed=9))
    assert [f.shape[1] for f in feats] == [64, 128, 256, 512]
    for i, f in enumerate(feats):
        np.testing.assert_allclose(f.numpy(), gold["relu%d" % i], **TOL)


def test_ssim_known_answers(manifest):
    a1, a2 = det_input((2, 3, 64, 64), seed=1), det_input((2, 3, 64, 64), seed=2)
    kat = manifest["kat"]["pytorch_ssim"]
    assert abs(ssim_ref.ssim(a1, a2).item() - kat["rand"][0]) < 1e-6
    assert abs(ssim_ref.ssim(a1, a1 * 0.9 + 0.1 * a2).item() - kat["near"][0]) < 1e-6
    assert ssim_ref.ssim(a1, a1).item() == pytest.approx(1.0, abs=1e-6)


def test_freqsplit_known_answers(golden_dir, manifest):
    k = freqsplit_ref.isotropic_gaussian_kernel(15, 3.0)
    assert abs(k.sum() - 1.0) < 1e-12
    assert abs(k[7, 7] - 0.0181167153) < 1e-9 and abs(k[0, 0] - 7.8268549e-05) < 1e-12
    c = torch.full((1, 3, 20, 20), 0.7)
    assert (freqsplit_ref.blur(c, use_input_norm=False) - 0.7).abs().max() < 1e-6
    lc = freqsplit_ref.laplacian(torch.ones(1, 3, 8, 8))
    assert lc[0, 0, 3, 3] == 0 and lc[0, 0, 0, 0] == -5 and lc[0, 1, 0, 3] == -3
    with pytest.raises(ValueError):
        freqsplit_ref.laplacian(torch.ones(3, 8, 8))
    gold = _load(golden_dir, "freqsplit.npz")
    x = det_input((2, 3, 40, 48), seed=11)
    np.testing.assert_allclose(freqsplit_ref.blur(x, use_input_norm=True).numpy(), gold["blur_norm"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(freqsplit_ref.laplacian(x).numpy(), gold["lap"], rtol=1e-4, atol=1e-5)


def test_dehaze22_d_matches_golden(golden_dir, manifest):
    """dehaze22.D restatement vs the output the REAL reference produced (make_golden.py)."""
    from oracle import dehaze22_ref as o22
    d = o22.D(9, 36)
    sd = fill_state_dict(d, seed=2)
    assert list(sd.keys()) == manifest["d22_keys"]
    x = det_input((2, 9, 64, 64), seed=77, lo=-1.0, hi=1.0)
    with torch.no_grad():
        y = d(x)
    assert y.shape == (2, 1, 6, 6)
    np.testing.assert_allclose(y.numpy(), _load(golden_dir, "d22_2x64.npz")["y"], **TOL)


def test_oracle_reproduces_wellconditioned_reference_gradients(golden_dir):
    """The 282-parameter gradient fixture (reference run of oracle/make_golden.py, batch 8 @ 64x64, BatchNorm biases + 3):
    the oracle restatement reproduces every stored projection (same torch CPU ops: exact up to reduction order)."""
    import json
    from oracle import dehaze1113_ref as ref
    from oracle.detweights import det_input, fill_state_dict, grad_projection, shift_bn_bias
    gold = _load(golden_dir, "fdgan_8x64_wellcond.npz")
    man = json.load(open(os.path.join(golden_dir, "MANIFEST.json")))["fdgan_wellcond"]
    og = ref.FDGAN()
    fill_state_dict(og, seed=0)
    shift_bn_bias(og, man["bn_bias_shift"])
    x, tgt = det_input((8, 3, 64, 64), seed=1234), det_input((8, 3, 64, 64), seed=4321, lo=-1.0, hi=1.0)
    y = og(x)
    ((y - tgt) ** 2).mean().backward()
    np.testing.assert_allclose(y.detach().numpy()[:, :, ::4, ::4], gold["y"], **TOL)
    n = 0
    for name, p in og.named_parameters():
        key = "proj__" + name.replace(".", "__")
        if key not in gold.files:
            assert p.grad is None
            continue
        proj, norm = grad_projection(name, p.grad.numpy())
        scale = float(np.abs(gold[key]).max()) + 1e-12
        assert np.abs(proj - gold[key]).max() < 1e-4 * scale + 1e-9, name
        n += 1
    assert n == 282


def test_wellconditioned_tolerance_rows_match_the_committed_emulation(golden_dir):
    """The per-parameter tolerance rows `o2o__*` of fdgan_8x64_wellcond.npz (and MANIFEST's oracle_vs_emulated_median) are the
    distance between the fp32 oracle and tests/hiputil.emulate_kernel_operands AS COMMITTED.  VERDICT r3 weak #1: the helper
    moved from bf16 to fp16 rounding while the fixture kept the bf16-era rows (median 3.2 % where the recipe gives 0.8 %), so the
    GPU test's bound was 4x looser than its docstring.  This re-runs the emulation here and fails when the fixture is stale."""
    import json
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from hiputil import emulate_kernel_operands, rel_rms
    from oracle import dehaze1113_ref as ref
    from oracle.detweights import det_input, fill_state_dict, shift_bn_bias
    gold = _load(golden_dir, "fdgan_8x64_wellcond.npz")
    man = json.load(open(os.path.join(golden_dir, "MANIFEST.json")))["fdgan_wellcond"]
    og, oe = ref.FDGAN(), ref.FDGAN()
    fill_state_dict(og, seed=0)
    shift_bn_bias(og, man["bn_bias_shift"])
    oe.load_state_dict(og.state_dict())
    emulate_kernel_operands(oe, round_grads=True)
    x, tgt = det_input((8, 3, 64, 64), seed=1234), det_input((8, 3, 64, 64), seed=4321, lo=-1.0, hi=1.0)
    ((og(x.clone()) - tgt) ** 2).mean().backward()
    ((oe(x.clone()) - tgt) ** 2).mean().backward()
    pe = dict(oe.named_parameters())
    now, then = [], []
    for name, p in og.named_parameters():
        key = "o2o__" + name.replace(".", "__")
        if key not in gold.files:
            continue
        now.append(rel_rms(pe[name].grad, p.grad))
        then.append(float(gold[key]))
    now, then = np.array(now), np.array(then)
    assert len(now) == 282
    # rounding decisions are deterministic; only fp32 reduction order (thread count) moves a row, by far less than this
    assert abs(np.median(now) - man["oracle_vs_emulated_median"]) < 0.1 * man["oracle_vs_emulated_median"], (np.median(now), man)
    assert abs(np.median(then) - man["oracle_vs_emulated_median"]) < 1e-9
    ok = np.abs(now - then) <= 0.25 * then + 1e-4
    assert ok.mean() > 0.97, (float(ok.mean()), float(np.median(now)), float(np.median(then)))


def test_loss_module_restatements_are_pinned_to_the_reference_bytecode(golden_dir):
    """tests/golden/loss_pyc_constants.json = names / constants / line numbers extracted from the reference's
    __pycache__/loss.cpython-36.pyc (oracle/pin_loss_pyc.py); the Blur / Laplacian / ContextualLoss restatements must
    carry exactly those constants, defaults and call structure."""
    import inspect
    import json
    from oracle import contextual_ref
    t = json.load(open(os.path.join(golden_dir, "loss_pyc_constants.json")))
    assert t["__header__"]["magic"] == 3379 and t["__header__"]["source"].endswith("loss.py")
    mod = t["<module>"]
    # blur_kernel = isotropic_gaussian_kernel(l=15, sigma=3.0); blur = Blur(l=15, kernel=blur_kernel); laplace_filter = Laplacian(3)
    for name in ("blur_kernel", "blur", "laplace_filter", "isotropic_gaussian_kernel", "Blur", "Laplacian", "ContextualLoss"):
        assert name in mod["names"], name
    assert 15 in mod["consts"] and 3.0 in mod["consts"] and ["l", "sigma"] in mod["consts"] and ["l", "kernel"] in mod["consts"]
    assert 3 in mod["consts"] and ["kernel_size"] in mod["consts"]
    sig = inspect.signature(freqsplit_ref.blur)
    assert sig.parameters["l"].default == 15 and sig.parameters["sigma"].default == 3.0 and sig.parameters["use_input_norm"].default is True
    # Blur.__init__(self, l=15, kernel=None, use_input_norm=True): ReflectionPad2d(l // 2), ImageNet mean / std buffers
    assert t["Blur"]["consts"].count(15) >= 1 and [15, None, True] in t["Blur"]["consts"]
    bi = t["Blur.__init__"]
    assert bi["args"] == ["self", "l", "kernel", "use_input_norm"] and "ReflectionPad2d" in bi["names"] and 2 in bi["consts"]
    assert [c for c in bi["consts"] if isinstance(c, float)] == list(freqsplit_ref.IMAGENET_MEAN) + list(freqsplit_ref.IMAGENET_STD)
    assert "conv2d" in t["Blur.forward"]["names"] and "pad" in t["Blur.forward"]["names"]
    ig = t["isotropic_gaussian_kernel"]
    assert ig["args"] == ["l", "sigma", "tensor"] and {"arange", "meshgrid", "exp", "sum"} <= set(ig["names"]) and 2.0 in ig["consts"]
    # Laplacian: ones(k, k) with centre 1 - k^2, conv2d(padding=, stride=, groups=), padding (k - 1) // 2
    assert "ones" in t["get_laplacian_kernel2d"]["names"] and 2 in t["get_laplacian_kernel2d"]["consts"]
    lf = t["Laplacian.forward"]
    assert ["padding", "stride", "groups"] in lf["consts"] and "repeat" in lf["names"] and 4 in lf["consts"]
    assert t["Laplacian.compute_zero_padding"]["consts"][1:] == [1, 2]
    assert float(freqsplit_ref.laplacian_kernel2d(3)[1, 1]) == -8.0 and float(freqsplit_ref.laplacian_kernel2d(5)[2, 2]) == -24.0
    # ContextualLoss(sigma=0.1, b=1.0, epsilon=1e-5, similarity='cos') and its methods, at the bytecode's line numbers
    cl = t["ContextualLoss"]
    assert [0.1, 1.0, 1e-05, "cos"] in cl["consts"]
    d = inspect.signature(contextual_ref.ContextualLoss.__init__).parameters
    assert [d[k].default for k in ("sigma", "b", "epsilon", "similarity")] == [0.1, 1.0, 1e-05, "cos"]
    lines = {"__init__": 24, "cos_similarity": 31, "L2_similarity": 46, "relative_distances": 49,
             "weighted_average_distances": 53, "CX": 59, "forward": 70}
    for meth, line in lines.items():
        assert t["ContextualLoss." + meth]["line"] == line and hasattr(contextual_ref.ContextualLoss, meth)
    assert {"mean", "sqrt", "sum", "bmm", "permute", "div"} <= set(t["ContextualLoss.cos_similarity"]["names"])
    assert {"min", "e"} <= set(t["ContextualLoss.relative_distances"]["names"])
    assert {"exp", "b", "sigma", "sum", "div"} <= set(t["ContextualLoss.weighted_average_distances"]["names"])
    assert {"max", "mean", "log", "permute"} <= set(t["ContextualLoss.CX"]["names"])
    # a known answer: identical features -> every row's best match is itself with distance 0 -> CX close to its maximum
    torch.manual_seed(0)
    f = torch.rand(2, 8, 5, 5)
    same, other = contextual_ref.ContextualLoss()(f, f.clone()), contextual_ref.ContextualLoss()(f, torch.rand(2, 8, 5, 5))
    assert float(same) < 1e-3 and float(other) > float(same) + 0.1


@pytest.mark.parametrize("kind", ["G", "G2"])
def test_legacy_unets_match_golden(golden_dir, manifest, kind):
    """oracle/legacy_ref.py (dehaze22.G :205-362, G2 :364-488) against outputs of the REAL reference --
    eval mode, and train mode with the Dropout2d masks the reference drew (stored with the fixture)."""
    from oracle import legacy_ref
    import models.dehaze22 as net22
    assert manifest["ref_vs_oracle_maxabs"]["legacy_%s_eval" % kind] == 0.0
    assert manifest["ref_vs_oracle_maxabs"]["legacy_%s_train" % kind] == 0.0
    net = getattr(net22, kind)(3, 3, 8)                     # the product's parameter container: same keys, same shapes
    assert list(net.state_dict().keys()) == manifest["legacy_%s_keys" % kind]
    fill_state_dict(net, seed=5)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    if kind == "G":
        sd["dlayerfinal.dlayer1.conv.weight"] *= 0.3
    x = det_input((2, 3, 256, 256), seed=21)
    g = _load(golden_dir, "legacy_%s_2x256.npz" % kind.lower())
    with torch.no_grad():
        ye, _ = legacy_ref.unet_forward({k: v.clone() for k, v in sd.items()}, x.clone(), False, kind)
        sdt = {k: v.clone() for k, v in sd.items()}
        yt, used = legacy_ref.unet_forward(sdt, x.clone(), True, kind, masks=list(torch.from_numpy(g["masks"])))
    assert float((ye[:, :, ::4, ::4] - torch.from_numpy(g["y_eval"])).abs().max()) < 1e-6
    assert float((yt[:, :, ::4, ::4] - torch.from_numpy(g["y_train"])).abs().max()) < 1e-5
    assert float((sdt["dlayer5.dlayer5.bn.running_mean"] - torch.from_numpy(g["rm_dlayer5"])).abs().max()) < 1e-6
    assert float((sdt["layer8.layer8.bn.running_var"] - torch.from_numpy(g["rv_layer8"])).abs().max()) < 1e-6
    assert int(sdt["layer8.layer8.bn.num_batches_tracked"]) == 1


def test_transposed_conv_as_four_parity_convolutions():
    """models/dehaze22.py expresses ConvTranspose2d(4, 2, 1) as four stride-1 3x3 pad-1 convolutions, one per output parity:
    the filter rearrangement against torch's own conv_transpose2d."""
    import torch.nn.functional as F
    import models.dehaze22 as net22
    torch.manual_seed(3)
    w = torch.randn(5, 7, 4, 4)
    x = torch.randn(2, 5, 6, 9)
    ref = F.conv_transpose2d(x, w, None, 2, 1)
    filt = net22._phase_filters(w, torch.empty(4, 7, 5, 3, 3))
    out = torch.zeros_like(ref)
    for a in range(2):
        for b in range(2):
            out[:, :, a::2, b::2] = F.conv2d(x, filt[a * 2 + b], None, 1, 1)
    assert float((out - ref).abs().max()) < 1e-5


_DENSE = (("dense1113", "dehaze1113", "Dense", "bn"), ("dense2_1113", "dehaze1113", "Dense2", "pyramid"), ("dense22", "dehaze22", "Dense", "pyramid"))


@pytest.mark.parametrize("nm,mod,cls,tail", _DENSE)
def test_legacy_dense_matches_golden(golden_dir, manifest, nm, mod, cls, tail):
    """oracle/legacy_ref.dense_forward (dehaze1113.Dense :431-570, Dense2 :572-699, dehaze22.Dense :531-660)
    against outputs of the REAL reference, eval and train mode, incl. the running statistics a train-mode forward leaves."""
    import importlib
    from oracle import legacy_ref
    assert manifest["ref_vs_oracle_maxabs"]["legacy_%s_eval" % nm] == 0.0 and manifest["ref_vs_oracle_maxabs"]["legacy_%s_train" % nm] == 0.0
    net = getattr(importlib.import_module("models." + mod), cls)()          # the product's parameter container
    assert len(net.state_dict()) == manifest["legacy_%s_keys" % nm]
    fill_state_dict(net, seed=6)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    sd["refine3.weight"] *= 0.1
    sd["refine3.bias"] *= 0.1
    x = det_input((2, 3, 128, 160), seed=31)
    g = _load(golden_dir, "legacy_%s_2x128.npz" % nm)
    with torch.no_grad():
        ye = legacy_ref.dense_forward({k: v.clone() for k, v in sd.items()}, x.clone(), False, tail)
        sdt = {k: v.clone() for k, v in sd.items()}
        yt = legacy_ref.dense_forward(sdt, x.clone(), True, tail)
    assert float((ye[:, :, ::2, ::2] - torch.from_numpy(g["y_eval"])).abs().max()) < 1e-5
    assert float((yt[:, :, ::2, ::2] - torch.from_numpy(g["y_train"])).abs().max()) < 1e-5
    assert float((sdt["norm0.running_var"] - torch.from_numpy(g["rv_norm0"])).abs().max()) < 1e-5
    assert float((sdt["trans_block5.bn1.running_mean"] - torch.from_numpy(g["rm_tb5"])).abs().max()) < 1e-5


def test_densenet_stem_as_space_to_depth_convolution():
    """models/dehaze1113.py runs conv0 (7x7, stride 2, pad 3) as a 4x4 stride-1 convolution over the 2x2 space-to-depth image
    with two zero block rows / columns in front and one behind: the filter rearrangement against torch's conv2d."""
    import torch.nn.functional as F
    import models.dehaze1113 as net
    torch.manual_seed(4)
    w7 = torch.randn(8, 3, 7, 7)
    x = torch.randn(2, 3, 20, 28)
    ref = F.conv2d(x, w7, None, 2, 3)
    w4 = net._stem_filter(w7, torch.empty(8, 16, 4, 4))
    xs = torch.zeros(2, 16, 10 + 3, 14 + 3)
    xs[:, :12, 2:12, 2:16] = F.pixel_unshuffle(x, 2)
    out = F.conv2d(xs, w4)
    assert out.shape == ref.shape and float((out - ref).abs().max()) < 1e-4


def test_legacy_dehaze_matches_golden(golden_dir, manifest):
    """oracle/legacy_ref.dehaze_forward (dehaze22.py:662-753) against the four outputs of the REAL reference, eval mode (the
    train-mode pass, 0.0 from the reference as well, is left to the manifest: it costs another 20 s of CPU)."""
    from oracle import legacy_ref
    import models.dehaze22 as net22
    assert manifest["ref_vs_oracle_maxabs"]["legacy_dehaze_eval"] == 0.0 and manifest["ref_vs_oracle_maxabs"]["legacy_dehaze_train"] == 0.0
    net = net22.dehaze(3, 3, 64)
    assert len(net.state_dict()) == manifest["legacy_dehaze_keys"]
    fill_state_dict(net, seed=8)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    sd["tran_dense.refine3.weight"] *= 0.05
    sd["tran_dense.refine3.bias"].fill_(1.0)
    sd["refine3.weight"] *= 0.02
    x = det_input((2, 3, 256, 256), seed=41)
    g = _load(golden_dir, "legacy_dehaze_2x256.npz")
    with torch.no_grad():
        yo = legacy_ref.dehaze_forward(sd, x.clone(), False)
    for nm, t in zip(("dehaze", "tran", "atp", "dehaze2"), yo[:4]):
        assert float((t[:, :, ::8, ::8] - torch.from_numpy(g[nm + "_eval"])).abs().max()) < 1e-4, nm


def test_oracle_training_step_learns_on_a_fixed_batch():
    """VERDICT r4 next #7 (b), oracle side: the loss composition of the training step is this repository's reconstruction (the
    reference ships no training loop), so nothing pins it except that it TRAINS.  Thirty oracle steps on one fixed batch
    (2 x 64 x 64): L1 falls by more than 30 %, SSIM rises, every term stays finite.  (The HIP path is held to the same, and to the
    oracle's trajectory, in tests/test_hip_models.py::test_training_trajectory_matches_oracle_and_learns.)"""
    import math
    import torch
    from oracle.train_ref import TrainStepRef
    from oracle.detweights import det_input
    torch.manual_seed(3)
    ref = TrainStepRef()
    gt = det_input((2, 3, 64, 64), seed=5)
    haze = (gt * 0.6 + 0.3).clamp(0, 1)
    traj = [ref.step(haze, gt) for _ in range(30)]
    assert all(math.isfinite(v) for r in traj for v in r.values())
    assert traj[-1]["l1"] < 0.7 * traj[0]["l1"], (traj[0]["l1"], traj[-1]["l1"])
    assert traj[-1]["ssim"] > traj[0]["ssim"] + 0.1, (traj[0]["ssim"], traj[-1]["ssim"])
    assert min(r["l1"] for r in traj[10:]) < 0.6 * traj[0]["l1"]
