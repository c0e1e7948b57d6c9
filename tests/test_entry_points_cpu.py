"""CPU tests of the tiers either side of the hot path: PSNR/SSIM scorer (known answers from the
reference's own functions, tests/golden/MANIFEST.json), dataset indexing, image saving, misc."""
import os

import numpy as np
import pytest
import torch


def test_psnrssim_known_answer(manifest):
    import PSNRSSIM as ps
    rng = np.random.default_rng(0)
    a = rng.integers(0, 256, (64, 80, 3), dtype=np.uint8)
    b = np.clip(a.astype(int) + rng.integers(-20, 21, a.shape), 0, 255).astype(np.uint8)
    ssim = np.mean([ps.compare_ssim(a[..., i], b[..., i]) for i in range(3)])
    kat = manifest["kat"]["psnrssim"]
    assert abs(ssim - kat["ssim"]) < 1e-12                      # 0.9875573750475048
    assert abs(ps.psnr_images(a, b) - kat["psnr"]) < 1e-9       # 26.8562966261433 after the 1-px strip
    assert ps.compare_ssim(a[..., 0], a[..., 0]) == pytest.approx(1.0)
    with pytest.raises(ValueError):
        ps.compare_ssim(a[..., 0], b[:32, :, 0])


def test_psnrssim_cli_on_png_dirs(tmp_path, capsys):
    import PSNRSSIM as ps
    from PIL import Image
    rng = np.random.default_rng(1)
    g, r = tmp_path / "gt", tmp_path / "res"
    g.mkdir(), r.mkdir()
    for i in (10, 2):                                            # string sort: '10.png' < '2.png'
        a = rng.integers(0, 256, (40, 48, 3), dtype=np.uint8)
        b = np.clip(a.astype(int) + rng.integers(-9, 10, a.shape), 0, 255).astype(np.uint8)
        Image.fromarray(a).save(g / ("%d.png" % i))
        Image.fromarray(b).save(r / ("%d.png" % i))
    psnr, ssim = ps.main(["--gt_dir", str(g), "--result_dir", str(r)])
    out = capsys.readouterr().out
    assert out.index("10.png") < out.index("2.png") and "compute ssim" in out
    assert 30 < psnr < 40 and 0.9 < ssim < 1.0
    assert round(psnr, 4) == psnr and round(ssim, 4) == ssim     # 4-decimal quantisation


def test_pix2pix_dataset_indexing(tmp_path):
    from datasets.pix2pix import pix2pix, write_pair
    rng = np.random.default_rng(1234)
    root = str(tmp_path / "ds")
    pairs = []
    for i in range(2):
        haze, gt = rng.random((24, 32, 3)), rng.random((24, 32, 3))
        path = write_pair(root, i, haze, gt)
        assert os.path.basename(path).split(".")[0] == str(i)
        pairs.append((np.float32(haze), np.float32(gt)))
    ds = pix2pix(root, transform=object(), seed=3)
    assert len(ds) == 2
    for i in range(2):
        haze, gt = ds[i]
        assert haze.shape == (3, 24, 32) and haze.dtype == np.float32
        np.testing.assert_array_equal(haze, pairs[i][0].transpose(2, 0, 1))   # bit-exact
        np.testing.assert_array_equal(gt, pairs[i][1].transpose(2, 0, 1))
    with pytest.raises(FileNotFoundError):
        ds[2]                                                    # items are <index>.h5, not sorted names
    assert len(pix2pix(str(tmp_path / "empty"))) == 0


def test_loader_and_helpers(tmp_path):
    import misc
    from datasets.pix2pix import write_pair
    root = str(tmp_path / "ds")
    for i in range(3):
        write_pair(root, i, np.full((8, 8, 3), i / 4.0), np.zeros((8, 8, 3)))
    loader = misc.getLoader('pix2pix', root, 1024, 1024, 1, 0, split='Train', shuffle=False, seed=None)
    got = [float(h[0, 0, 0, 0]) for h, _ in loader]
    assert got == [0.0, 0.25, 0.5]
    with pytest.raises(ValueError):
        misc.getLoader('pix2pix_val2', root, 1, 1)
    m = misc.AverageMeter()
    m.update(2.0, 3), m.update(4.0, 1)
    assert m.avg == 2.5 and m.count == 4
    m.avg = 7.0                                  # a plain attribute, as in the reference (misc.py:121-136)
    m.reset()
    assert (m.val, m.avg, m.sum, m.count) == (0, 0, 0, 0)
    opt = torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))], lr=2e-4)
    misc.adjust_learning_rate(opt, 2e-4, 0, 0, 4)
    assert opt.param_groups[0]['lr'] == pytest.approx(1.5e-4)
    for _ in range(5):
        misc.adjust_learning_rate(opt, 2e-4, 0, 0, 4)
    assert opt.param_groups[0]['lr'] == 0
    pool = misc.ImagePool(2)
    a, b = torch.zeros(1), torch.ones(1)
    assert pool.query(a) is a and pool.query(b) is b and pool.num_imgs == 2
    # full pool: a fair coin between "return the new image" and "swap with a uniformly drawn slot, return the old one"
    pool = misc.ImagePool(4, seed=11)
    for i in range(4):
        pool.query(torch.full((2,), float(i)))
    kept, swapped, seen = 0, 0, set()
    for i in range(4, 404):
        new = torch.full((2,), float(i))
        got = pool.query(new)
        if got is new:
            kept += 1
        else:
            swapped += 1
            assert float(got[0]) < i and float(got[0]) not in seen          # an older image, returned once
            seen.add(float(got[0]))
            assert any(float(pool.images[s][0]) == i for s in range(4))     # the new one took its slot
    assert 150 < kept < 250 and kept + swapped == 400
    assert misc.ImagePool(0).query(a) is a
    # queries of different shapes (a ragged last batch, another image size): stored per slot, never broadcast
    pool = misc.ImagePool(2, seed=3)
    pool.query(torch.zeros(4, 3, 8, 8)), pool.query(torch.ones(4, 3, 8, 8))
    outs = [pool.query(torch.full(shape, 2.0 + i)) for i, shape in enumerate([(1, 3, 8, 8), (4, 3, 16, 16), (4, 3, 8, 8)] * 8)]
    assert all(o.dim() == 4 for o in outs) and {tuple(t.shape) for t in pool.images} <= {(4, 3, 8, 8), (1, 3, 8, 8), (4, 3, 16, 16)}
    assert all(bool((t == t.flatten()[0]).all()) for t in pool.images)       # every slot holds ONE whole image batch
    np.random.seed(5)
    r1 = misc.ImagePool(3).rng.integers(1 << 30)
    np.random.seed(5)
    assert misc.ImagePool(3).rng.integers(1 << 30) == r1                     # np.random.seed() still pins a run
    conv, bn = torch.nn.ConvTranspose2d(4, 4, 1), torch.nn.BatchNorm2d(4)
    torch.manual_seed(0)
    misc.weights_init(conv), misc.weights_init(bn)
    assert conv.weight.std() < 0.05 and abs(float(bn.weight.mean()) - 1.0) < 0.05 and float(bn.bias.abs().sum()) == 0


def test_save_image_minmax_normalisation(tmp_path):
    import misc
    from PIL import Image
    t = torch.tensor([[[-1.0, 0.0], [0.5, 1.0]]]).repeat(3, 1, 1)
    arr = misc.to_uint8_image(t)
    assert arr.shape == (2, 2, 3) and arr.dtype == np.uint8
    assert arr[0, 0, 0] == 0 and arr[1, 1, 0] == 255 and arr[0, 1, 0] == 128 and arr[1, 0, 0] == 191
    p = str(tmp_path / "x.png")
    misc.save_image(t, p, normalize=True)
    np.testing.assert_array_equal(np.asarray(Image.open(p)), arr)


def test_demo_checkpoint_key_handling(tmp_path):
    import demo
    sd = {"module.conv_refin1.weight": torch.zeros(1), "module.dense_block1.denselayer1.norm.1.weight": torch.ones(1),
          "module.dense_block1.denselayer1.conv.2.weight": torch.ones(1), "trans_block1.norm.weight": torch.ones(1)}
    p = str(tmp_path / "g.pth")
    torch.save(sd, p)
    out = demo.load_generator_state(p)
    assert list(out) == ["conv_refin1.weight", "dense_block1.denselayer1.norm1.weight",
                         "dense_block1.denselayer1.conv2.weight", "trans_block1.norm.weight"]
    opt = demo.build_parser().parse_args(["--valDataroot", "d", "--netG", "w.pth"])
    assert (opt.valBatchSize, opt.imageSize, opt.lrG, opt.beta1, opt.dataset) == (1, 1024, 0.0002, 0.5, 'pix2pix')


def test_transposed_conv_parity_filters_and_their_adjoint():
    """models/dehaze22.py: ConvTranspose2d(4, 2, 1) as four stride-1 3x3 convolutions, one per output parity (`_phase_filters`), and
    the map that gathers the four filter gradients back into the (cin, cout, 4, 4) parameter (`_phase_filter_grads`): the parity
    convolutions reproduce F.conv_transpose2d exactly, and the gather is the exact adjoint of the scatter (host logic, no GPU)."""
    import torch.nn.functional as F
    from models.dehaze22 import _phase_filter_grads, _phase_filters
    g = torch.Generator().manual_seed(3)
    cin, cout = 5, 7
    w = torch.randn(cin, cout, 4, 4, generator=g)
    x = torch.randn(2, cin, 6, 9, generator=g)
    filt = _phase_filters(w, [torch.zeros(cout, cin, 3, 3) for _ in range(4)])
    want = F.conv_transpose2d(x, w, None, 2, 1)
    got = torch.zeros_like(want)
    for a in range(2):
        for b in range(2):
            got[:, :, a::2, b::2] = F.conv2d(x, filt[a * 2 + b], None, 1, 1)
    assert float((got - want).abs().max()) < 1e-5
    d = [torch.randn(cout, cin, 3, 3, generator=g) for _ in range(4)]
    lhs = sum(float((f * dd).sum()) for f, dd in zip(filt, d))
    rhs = float((w * _phase_filter_grads(d, w)).sum())
    assert abs(lhs - rhs) < 1e-4 * max(1.0, abs(lhs))
    d[2] = None                                   # a parity that received no gradient counts as zero
    assert _phase_filter_grads(d, w).shape == w.shape


def test_stem_filter_and_its_gradient_map():
    """models/dehaze1113.py: DenseNet's 7x7 stride-2 stem as a 4x4 stride-1 convolution on the 2x2 space-to-depth image (`_stem_filter`),
    and `_DenseBase._derived_grads`' inverse map for its gradient: same outputs as F.conv2d(x, w7, stride 2, pad 3), and the gradient
    map is the adjoint."""
    import torch.nn.functional as F
    from models.dehaze1113 import _stem_filter
    g = torch.Generator().manual_seed(4)
    w7 = torch.randn(8, 3, 7, 7, generator=g)
    x = torch.randn(2, 3, 16, 24, generator=g)
    w4 = _stem_filter(w7, torch.zeros(8, 16, 4, 4))
    xs = F.pad(F.pixel_unshuffle(x, 2), (2, 1, 2, 1))           # two zero block columns / rows in front, one behind
    xs = torch.cat([xs, torch.zeros(2, 4, xs.shape[2], xs.shape[3])], 1)
    assert float((F.conv2d(xs, w4) - F.conv2d(x, w7, None, 2, 3)).abs().max()) < 1e-4
    d4 = torch.randn(8, 16, 4, 4, generator=g)
    g8 = torch.zeros(8, 3, 8, 8)
    for dy in range(2):
        for dx in range(2):
            g8[:, :, dy::2, dx::2] = d4[:, dy * 2 + dx:12:4]
    assert abs(float((w4 * d4).sum()) - float((w7 * g8[:, :, 1:, 1:]).sum())) < 1e-4 * max(1.0, abs(float((w4 * d4).sum())))


@pytest.mark.parametrize("kind", ["G", "G2"])
def test_unet_table_gradients_reach_their_modules(kind):
    """models/dehaze22.py `_UNet._table_grads` (host logic, no GPU): the dgamma / dbeta of a concat buffer's side-by-side norm table go to
    the two BatchNorm modules the table was built from -- left half dlayer k+1's norm, right half layer k's.  Train mode: dlayer 6 / 7
    are dropout levels whose norm is applied (and differentiated) by the bn_dropout record, so their table entries are identities and
    must NOT receive a table gradient; eval mode (round 6): no dropout, all of dlayer 2..7's norms are table entries.  dlayer8 and
    layer1 have no norm in either mode."""
    import types
    import models.dehaze22 as net22
    net = getattr(net22, kind)(3, 3, 8)
    cd, ce = net.cd, net.ce
    for training in (True, False):
        net.train(training)
        tab = {k: dict(gamma=torch.zeros(cd[k + 1] + ce[k - 1]), beta=torch.zeros(cd[k + 1] + ce[k - 1])) for k in range(1, 8)}
        grads = {}
        for k in range(1, 8):
            grads[tab[k]["gamma"]] = torch.full((cd[k + 1] + ce[k - 1],), float(k))
            grads[tab[k]["beta"]] = torch.full((cd[k + 1] + ce[k - 1],), float(-k))
        net._table_grads(types.SimpleNamespace(tab=tab), grads)
        assert all(t["gamma"] not in grads and t["beta"] not in grads for t in tab.values())      # the table leaves are consumed
        got = {id(p): g for p, g in grads.items()}
        for j in range(2, 9):                 # encoder norm of layer j: right half of table j (j <= 7); layer8's norm is not in a table
            bn = getattr(net, "layer%d" % j)[0].bn
            if j <= 7:
                assert torch.equal(got[id(bn.weight)], torch.full((ce[j - 1],), float(j))) and torch.equal(got[id(bn.bias)], torch.full((ce[j - 1],), float(-j)))
            else:
                assert id(bn.weight) not in got
        for j in range(2, 8):                 # decoder norm of dlayer j: left half of table j - 1
            bn = getattr(net, "dlayer%d" % j)[0].bn
            if j <= 5 or not training:
                assert torch.equal(got[id(bn.weight)], torch.full((cd[j],), float(j - 1))), (training, j)
                assert torch.equal(got[id(bn.bias)], torch.full((cd[j],), float(1 - j))), (training, j)
            else:
                assert id(bn.weight) not in got and id(bn.bias) not in got, (training, j)
