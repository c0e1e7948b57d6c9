"""TEST INFRASTRUCTURE (oracle).  Deterministic, torch-RNG-independent
parameter fill keyed by state_dict key name, so the build container and the
GPU box construct bit-identical weights without shipping checkpoints
(SURVEY section 7 step 1).  numpy PCG64 streams are platform-stable.
"""
import zlib

import numpy as np
import torch


def _rng(name, seed):
    return np.random.Generator(np.random.PCG64([seed, zlib.crc32(name.encode())]))


def fill_state_dict(module, seed=0):
    """Overwrites every entry of module.state_dict() in place and returns it.

    conv / convT weight : N(0, sqrt(2/fan_in)) so activations stay O(1)
    conv bias           : N(0, 0.05)
    BN weight           : U(0.5, 1.5);  BN bias: N(0, 0.1)
    running_mean        : N(0, 0.1);    running_var: U(0.5, 1.5)
    num_batches_tracked : 0
    """
    sd = module.state_dict()
    for name, t in sd.items():
        g = _rng(name, seed)
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            t.zero_()
            continue
        shape = tuple(t.shape)
        if t.dim() == 4:
            fan_in = shape[1] * shape[2] * shape[3]
            if "trans_block" in name and name.endswith("conv1.weight") and shape[2] == 1:
                fan_in = shape[0]          # ConvTranspose2d weight is (Cin, Cout, 1, 1)
            v = g.normal(0.0, np.sqrt(2.0 / fan_in), shape)
            if name.startswith("conv_refin3."):
                v = v * 0.3                # keep the tanh input O(0.5): unsaturated output
        elif leaf == "running_mean":
            v = g.normal(0.0, 0.1, shape)
        elif leaf == "running_var":
            v = g.uniform(0.5, 1.5, shape)
        elif leaf == "weight":              # BatchNorm gamma
            v = g.uniform(0.5, 1.5, shape)
        elif leaf == "bias":
            is_bn = (name.rsplit(".", 1)[0] + ".running_mean") in sd
            v = g.normal(0.0, 0.1 if is_bn else 0.05, shape)
        else:
            raise KeyError(name)
        t.copy_(torch.from_numpy(np.asarray(v, dtype=np.float32)))
    return sd


def det_input(shape, seed=1234, lo=0.0, hi=1.0):
    """U[lo,hi) float32 tensor from numpy.default_rng(seed)."""
    a = np.random.default_rng(seed).random(shape, dtype=np.float64) * (hi - lo) + lo
    return torch.from_numpy(a.astype(np.float32))
