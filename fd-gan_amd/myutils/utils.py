"""Weight loading for the perceptual-loss network behind the reference's `myutils.utils` interface
(/root/reference/myutils/utils.py:84-94).

    init_vgg16(model_folder)          converts <folder>/vgg16.t7 (the Lua-torch file of jcjohnson/fast-neural-style the
                                      reference downloads) to <folder>/vgg16.weight, a Vgg16 state_dict, by copying the
                                      Lua model's parameters POSITIONALLY onto Vgg16's (utils.py:90-94).  No download here
                                      (no network): a missing .t7 raises FileNotFoundError naming the URL.
    load_vgg16_weights(vgg, path)     loads any of: a Vgg16 state_dict file (`vgg16.weight`, keys conv1_1.weight ...),
                                      a torchvision-style VGG16 state_dict (features.<n>.weight / .bias: the first 13
                                      conv layers in order), or a `.t7`.

`torch.utils.serialization.load_lua`, which the reference calls, left PyTorch in 1.0; `read_t7` below is a reader for
the subset of the Torch7 binary serialisation such model files use (numbers, strings, booleans, tables, torch.*Tensor /
torch.*Storage objects, nn.* modules as attribute tables).
"""
import os
import struct

import numpy as np
import torch

VGG16_T7_URL = "http://cs.stanford.edu/people/jcjohns/fast-neural-style/models/vgg16.t7"
_TENSOR_DTYPES = {"Float": np.float32, "Double": np.float64, "Long": np.int64, "Int": np.int32, "Short": np.int16,
                  "Byte": np.uint8, "Char": np.int8}


class _T7Object(dict):
    """A deserialised torch class instance: attribute table + the Lua class name (e.g. 'nn.SpatialConvolution')."""

    def __init__(self, typename, fields):
        super().__init__(fields if isinstance(fields, dict) else {})
        self.typename = typename


class _T7Reader:
    TYPE_NIL, TYPE_NUMBER, TYPE_STRING, TYPE_TABLE, TYPE_TORCH, TYPE_BOOLEAN = 0, 1, 2, 3, 4, 5
    TYPE_FUNCTION, TYPE_RECUR_FUNCTION, LEGACY_TYPE_RECUR_FUNCTION = 6, 8, 7

    def __init__(self, f):
        self.f, self.memo = f, {}

    def _i32(self):
        return struct.unpack("<i", self.f.read(4))[0]

    def _i64(self):
        return struct.unpack("<q", self.f.read(8))[0]

    def _str(self):
        return self.f.read(self._i32()).decode("latin-1")

    def read(self):
        t = self._i32()
        if t == self.TYPE_NIL:
            return None
        if t == self.TYPE_NUMBER:
            v = struct.unpack("<d", self.f.read(8))[0]
            return int(v) if v == int(v) and abs(v) < 2 ** 53 else v
        if t == self.TYPE_BOOLEAN:
            return self._i32() == 1
        if t == self.TYPE_STRING:
            return self._str()
        if t in (self.TYPE_TABLE, self.TYPE_TORCH):
            idx = self._i32()
            if idx in self.memo:
                return self.memo[idx]
            if t == self.TYPE_TABLE:
                out = {}
                self.memo[idx] = out
                for _ in range(self._i32()):
                    k = self.read()
                    out[k] = self.read()
                return out
            version = self._str()
            cls = self._str() if version.startswith("V ") else version
            return self._torch_object(idx, cls)
        raise ValueError("t7: unsupported type tag %d (functions / userdata are not expected in a model file)" % t)

    def _torch_object(self, idx, cls):
        kind = cls.split(".")[-1]
        for prefix, dt in _TENSOR_DTYPES.items():
            if cls == "torch.%sTensor" % prefix:
                ndim = self._i32()
                size = [self._i64() for _ in range(ndim)]
                stride = [self._i64() for _ in range(ndim)]
                offset = self._i64() - 1
                storage = self.read()
                if storage is None or ndim == 0:
                    out = torch.empty(0, dtype=torch.from_numpy(np.zeros(0, dt)).dtype)
                else:
                    out = torch.as_strided(storage, size, stride, offset).clone()
                self.memo[idx] = out
                return out
            if cls == "torch.%sStorage" % prefix:
                n = self._i64()
                arr = np.frombuffer(self.f.read(n * np.dtype(dt).itemsize), dtype=dt).copy()
                out = torch.from_numpy(arr)
                self.memo[idx] = out
                return out
        obj = _T7Object(cls, {})
        self.memo[idx] = obj
        fields = self.read()               # nn.* modules serialise their attribute table
        if isinstance(fields, dict):
            obj.update(fields)
        return obj


def read_t7(path):
    """-> python structure of a Torch7 binary file: tables as dicts (1-based integer keys for arrays), tensors as torch
    tensors, class instances as dict-like `_T7Object`s with `.typename`."""
    with open(path, "rb") as f:
        return _T7Reader(f).read()


def t7_parameters(obj):
    """Lua `module:parameters()[1]`: weight then bias of every module, depth first in `modules` order."""
    out = []
    if isinstance(obj, dict):
        for k in ("weight", "bias"):
            if isinstance(obj.get(k), torch.Tensor) and obj[k].numel() > 0:
                out.append(obj[k])
        mods = obj.get("modules")
        if isinstance(mods, dict):
            for i in sorted(k for k in mods if isinstance(k, int)):
                out += t7_parameters(mods[i])
    return out


def _positional(vgg, tensors, what):
    dst = list(vgg.parameters())
    if len(tensors) < len(dst):
        raise ValueError("%s holds %d parameter tensors, Vgg16 needs %d (13 conv layers x weight, bias)" % (what, len(tensors), len(dst)))
    with torch.no_grad():
        for d, s in zip(dst, tensors):           # utils.py:92-93: `dst.data[:] = src`, positional, extra tensors ignored
            if tuple(d.shape) != tuple(s.shape):
                raise ValueError("%s: parameter shape %s does not match Vgg16's %s" % (what, tuple(s.shape), tuple(d.shape)))
            d.copy_(s.to(d.dtype))
    return vgg


def load_vgg16_weights(vgg, path):
    """Loads pretrained weights into a `myutils.vgg16.Vgg16` (on whatever device it lives) and marks it as loaded."""
    if path.endswith(".t7"):
        _positional(vgg, t7_parameters(read_t7(path)), path)
    else:
        sd = torch.load(path, map_location="cpu")
        if isinstance(sd, dict) and "state_dict" in sd:
            sd = sd["state_dict"]
        own = vgg.state_dict()
        if all(k in sd for k in own):
            vgg.load_state_dict({k: sd[k] for k in own})
        else:                                       # torchvision: features.0.weight, features.0.bias, features.2.weight, ...
            conv = sorted({int(k.split(".")[1]) for k in sd if k.startswith("features.") and k.endswith(".weight") and sd[k].dim() == 4})
            if len(conv) < 13:
                raise KeyError("%s is neither a Vgg16 state_dict (conv1_1.weight ...) nor a torchvision VGG16 (features.N.weight)" % path)
            _positional(vgg, [sd["features.%d.%s" % (i, wb)] for i in conv[:13] for wb in ("weight", "bias")], path)
    vgg.weights_loaded = True
    return vgg


def init_vgg16(model_folder):
    """utils.py:84-94: make sure <model_folder>/vgg16.weight exists, converting vgg16.t7 if that is what is there."""
    from myutils.vgg16 import Vgg16
    weight = os.path.join(model_folder, "vgg16.weight")
    if os.path.exists(weight):
        return weight
    t7 = os.path.join(model_folder, "vgg16.t7")
    if not os.path.exists(t7):
        raise FileNotFoundError("%s not found and there is no network to fetch it: download %s there (the reference's "
                                "init_vgg16 runs wget)" % (t7, VGG16_T7_URL))
    vgg = Vgg16()
    _positional(vgg, t7_parameters(read_t7(t7)), t7)
    torch.save(vgg.state_dict(), weight)
    return weight
