#!/usr/bin/env python
"""Writes the committed HDF5 fixture tests/golden/h5/{0,1}.h5 (+ expected.npz) ONCE, with datasets/h5lite.py.
Config 1 of SURVEY 8(d): float32 HWC `rng.random((H, W, 3))` with numpy.default_rng(1234); 32x40 keeps the files small."""
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "fd-gan_amd")]
from datasets.pix2pix import write_pair  # noqa: E402

out = os.path.join(ROOT, "tests", "golden", "h5")
rng = np.random.default_rng(1234)
exp, crc = {}, []
for i in range(2):
    haze, gt = rng.random((32, 40, 3)).astype(np.float32), rng.random((32, 40, 3)).astype(np.float32)
    p = write_pair(out, i, haze, gt)
    exp["haze%d" % i], exp["gt%d" % i] = haze, gt
    crc.append(zlib.crc32(open(p, "rb").read()))
np.savez(os.path.join(out, "expected.npz"), crc32=np.array(crc, dtype=np.int64), **exp)
print("wrote", out, crc)
