"""Pretrained-weight plumbing of the perceptual network and the encoder: myutils/utils.py
(init_vgg16 / load_vgg16_weights incl. a Torch7 `.t7` reader, /root/reference/myutils/utils.py:84-94) and
models/tv_densenet121.load_densenet121_weights.  No real checkpoint exists offline: the files are synthesised here in the
formats the loaders accept -- the `.t7` by a writer of the Torch7 serialisation (class-tagged objects, referenced tables,
1-based tensor storage offsets), the structure jcjohnson's vgg16.t7 has (nn.Sequential of nn.SpatialConvolution / nn.ReLU /
nn.SpatialMaxPooling)."""
import os
import struct
import warnings

import numpy as np
import pytest
import torch


class _T7Writer:
    def __init__(self):
        self.b, self.idx = bytearray(), 0

    def i32(self, v):
        self.b += struct.pack("<i", v)

    def string(self, s):
        self.i32(len(s))
        self.b += s.encode()

    def value(self, v):
        if v is None:
            self.i32(0)
        elif isinstance(v, bool):
            self.i32(5), self.i32(1 if v else 0)
        elif isinstance(v, (int, float)):
            self.i32(1)
            self.b += struct.pack("<d", float(v))
        elif isinstance(v, str):
            self.i32(2), self.string(v)
        elif isinstance(v, torch.Tensor):
            self.tensor(v)
        elif isinstance(v, tuple):                   # (class name, fields): a torch class instance
            self.i32(4)
            self.idx += 1
            self.i32(self.idx)
            self.string("V 1"), self.string(v[0])
            self.value(v[1])
        elif isinstance(v, dict):
            self.i32(3)
            self.idx += 1
            self.i32(self.idx)
            self.i32(len(v))
            for k, x in v.items():
                self.value(k), self.value(x)
        else:
            raise TypeError(type(v))

    def tensor(self, t):
        t = t.contiguous().float()
        self.i32(4)
        self.idx += 1
        self.i32(self.idx)
        self.string("V 1"), self.string("torch.FloatTensor")
        self.i32(t.dim())
        for s in t.shape:
            self.b += struct.pack("<q", s)
        for s in t.stride():
            self.b += struct.pack("<q", s)
        self.b += struct.pack("<q", 1)               # 1-based storage offset
        self.i32(4)
        self.idx += 1
        self.i32(self.idx)
        self.string("V 1"), self.string("torch.FloatStorage")
        self.b += struct.pack("<q", t.numel())
        self.b += t.numpy().tobytes()


def _lua_vgg16(params):
    """nn.Sequential like fast-neural-style's vgg16.t7: conv / relu pairs with pools, 13 convs (+ gradWeight fields)."""
    mods, k = {}, 1
    it = iter(params)
    for block in (2, 2, 3, 3, 3):
        for _ in range(block):
            w, b = next(it), next(it)
            mods[k] = ("nn.SpatialConvolution", {"weight": w, "bias": b, "gradWeight": torch.zeros(0), "kW": 3, "kH": 3, "train": False})
            mods[k + 1] = ("nn.ReLU", {"inplace": True, "threshold": 0})
            k += 2
        mods[k] = ("nn.SpatialMaxPooling", {"kW": 2, "kH": 2})
        k += 1
    return ("nn.Sequential", {"modules": mods, "train": False})


def test_vgg16_weight_loading_in_all_three_formats(tmp_path):
    from myutils import utils as U
    from myutils.vgg16 import Vgg16
    torch.manual_seed(3)
    src = Vgg16()
    params = [p.detach().clone() for p in src.parameters()]
    assert len(params) == 26
    # 1. Lua-torch .t7 -> init_vgg16 writes vgg16.weight (positional copy, utils.py:90-94)
    w = _T7Writer()
    w.value(_lua_vgg16(params))
    folder = str(tmp_path)
    open(os.path.join(folder, "vgg16.t7"), "wb").write(bytes(w.b))
    obj = U.read_t7(os.path.join(folder, "vgg16.t7"))
    assert obj.typename == "nn.Sequential" and obj["modules"][1].typename == "nn.SpatialConvolution"
    assert len(U.t7_parameters(obj)) == 26
    path = U.init_vgg16(folder)
    assert path.endswith("vgg16.weight") and U.init_vgg16(folder) == path            # second call: already there
    a = U.load_vgg16_weights(Vgg16(), path)
    assert a.weights_loaded and all(torch.equal(p, q) for p, q in zip(a.parameters(), params))
    b = U.load_vgg16_weights(Vgg16(), os.path.join(folder, "vgg16.t7"))
    assert all(torch.equal(p, q) for p, q in zip(b.parameters(), params))
    # 2. torchvision-style state_dict (features.N.weight / bias, plus a classifier that is ignored)
    tv, n = {}, 0
    for cfg in (2, 2, 3, 3, 3):
        for _ in range(cfg):
            tv["features.%d.weight" % n], tv["features.%d.bias" % n] = params[len(tv)], params[len(tv) + 1]
            n += 2
        n += 1
    tv["classifier.0.weight"] = torch.zeros(4, 4)
    torch.save(tv, os.path.join(folder, "tv.pth"))
    c = U.load_vgg16_weights(Vgg16(), os.path.join(folder, "tv.pth"))
    assert all(torch.equal(p, q) for p, q in zip(c.parameters(), params))
    # errors
    with pytest.raises(FileNotFoundError, match="vgg16.t7"):
        U.init_vgg16(str(tmp_path / "empty"))
    torch.save({"x": torch.zeros(1)}, os.path.join(folder, "bad.pth"))
    with pytest.raises(KeyError):
        U.load_vgg16_weights(Vgg16(), os.path.join(folder, "bad.pth"))


def test_densenet_encoder_weights_and_the_pretrained_warning(tmp_path):
    import models.dehaze1113 as net
    from models import tv_densenet121 as tv
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        g = net.FDGAN()                                       # calls densenet121(pretrained=True), dehaze1113.py:707
    assert any("randomly initialised" in str(w.message) for w in rec)
    torch.manual_seed(1)
    full = tv.DenseNet121()
    sd = {"features." + k: v for k, v in full.features.state_dict().items()}
    old = {k.replace("norm1.", "norm.1.").replace("conv1.", "conv.1.").replace("norm2.", "norm.2.").replace("conv2.", "conv.2."): v
           for k, v in sd.items()}                            # torchvision-0.2 spelling
    for name, d in (("new.pth", sd), ("old.pth", {"module." + k: v for k, v in old.items()})):
        torch.save(d, str(tmp_path / name))
        g2 = net.FDGAN.__new__(net.FDGAN)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            g2.__init__()
        tv.load_densenet121_weights(g2, str(tmp_path / name))
        assert g2.encoder_weights_loaded
        assert torch.equal(g2.dense_block2.denselayer7.conv2.weight, full.features.denseblock2.denselayer7.conv2.weight)
        assert torch.equal(g2.trans_block3.norm.running_var, full.features.transition3.norm.running_var)
        assert torch.equal(g2.conv0.weight, full.features.conv0.weight)
    torch.save({"features.conv0.weight": sd["features.conv0.weight"]}, str(tmp_path / "partial.pth"))
    with pytest.raises(KeyError, match="lacks"):
        tv.load_densenet121_weights(g, str(tmp_path / "partial.pth"))
