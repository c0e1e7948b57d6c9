"""Paired hazy / ground-truth dataset behind the reference's `datasets.pix2pix.pix2pix`
interface (/root/reference/datasets/pix2pix.py:38-166).

Semantics kept bit-exact: item `index` is the file `<root>/<index>.h5` holding float32
HWC arrays 'haze' and 'gt' in [0, 1] (written by generate_testsample.py:31-38); both are returned as
CHW via swapaxes(0,2), swapaxes(1,2) (== transpose(2,0,1)); `len` is the number of files matching
`<root>/*h5`; `transform` is accepted and ignored (reference :80-92 is commented out); `seed` only
seeds numpy (:49-50).

h5py is an optional dependency (absent from the build and GPU images): without it the files go through
`datasets/h5lite.py`, a self-contained reader / writer of exactly the HDF5 subset h5py's defaults produce for these
files (contiguous float32 datasets in the root group).  `<index>.npz` files with the same two keys are still
accepted when no `.h5` exists (older fixtures) -- same indexing rule, `len` = number of `*.npz`.
"""
import glob
import os

import numpy as np
import torch.utils.data as data

from . import h5lite

try:  # pragma: no cover - depends on the host image
    import h5py
except ImportError:  # the build / GPU images ship no h5py
    h5py = None


def _read_pair(root, index):
    stem = root + '/' + str(index)
    if os.path.exists(stem + '.h5'):
        with (h5py if h5py is not None else h5lite).File(stem + '.h5', 'r') as f:
            return f['haze'][:], f['gt'][:]
    if os.path.exists(stem + '.npz'):
        with np.load(stem + '.npz') as f:
            return f['haze'], f['gt']
    raise FileNotFoundError("no sample %s.h5 (or .npz): items are addressed by running index, "
                            "reference datasets/pix2pix.py:62" % stem)


def write_pair(root, index, haze, gt):
    """Writer with generate_testsample.py:31-38 semantics (float32 HWC in [0,1]): `<index>.h5` with datasets
    'gt' and 'haze', through h5py when it is installed, else through h5lite (same file structure)."""
    os.makedirs(root, exist_ok=True)
    haze, gt = np.float32(haze), np.float32(gt)
    stem = os.path.join(root, str(index))
    if h5py is not None:
        with h5py.File(stem + '.h5', 'w') as f:
            f.create_dataset('gt', data=gt)
            f.create_dataset('haze', data=haze)
        return stem + '.h5'
    return h5lite.write(stem + '.h5', {'gt': gt, 'haze': haze})


class pix2pix(data.Dataset):
    def __init__(self, root, transform=None, loader=None, seed=None):
        self.root = root
        self.transform = transform
        self.loader = loader
        if seed is not None:
            np.random.seed(seed)

    def __getitem__(self, index):
        haze_image, GT = _read_pair(self.root, index)
        haze_image = np.swapaxes(np.swapaxes(haze_image, 0, 2), 1, 2)
        GT = np.swapaxes(np.swapaxes(GT, 0, 2), 1, 2)
        return haze_image, GT

    def __len__(self):
        n = len(glob.glob(self.root + '/*h5'))
        return n if n else len(glob.glob(self.root + '/*npz'))
