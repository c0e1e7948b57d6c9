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
