"""Run helpers behind the reference's `misc` module (/root/reference/misc.py): loader factory and
the GAN-training utilities.  Semantics per SURVEY Appendix G; image saving follows
torchvision.utils.save_image(normalize=True) (demo.py:151), which the reference imports from
torchvision (absent here).
"""
import os

import numpy as np
import torch


def create_exp_dir(exp):
    """misc.py:7-13: make the experiment directory if it is not there yet; always True."""
    if not os.path.isdir(exp):
        os.makedirs(exp, exist_ok=True)
        print('Creating exp dir: %s' % exp)
    return True


def weights_init(m):
    """misc.py:16-22 semantics: dispatch on a substring of the class name -- 'Conv' (which also
    matches ConvTranspose2d) draws weight ~ N(0, 0.02) and leaves the bias alone; 'BatchNorm' draws weight ~ N(1, 0.02)
    and zeroes the bias.  For `net.apply(weights_init)`."""
    kind = type(m).__name__
    with torch.no_grad():
        if 'Conv' in kind:
            torch.nn.init.normal_(m.weight, mean=0.0, std=0.02)
        elif 'BatchNorm' in kind:
            torch.nn.init.normal_(m.weight, mean=1.0, std=0.02)
            torch.nn.init.zeros_(m.bias)


def getLoader(datasetName, dataroot, originalSize, imageSize, batchSize=64, workers=4,
              mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5), split='train', shuffle=True, seed=None):
    """misc.py:25-69.  originalSize / imageSize / mean / std are dead parameters in the reference
    (its transforms are commented out); only 'pix2pix' resolves to a working dataset."""
    if datasetName != 'pix2pix':
        raise ValueError("dataset %r: only 'pix2pix' works in the reference (misc.py:38 imports a class that "
                         "does not exist)" % datasetName)
    from datasets.pix2pix import pix2pix as commonDataset
    dataset = commonDataset(root=dataroot, transform=None, seed=seed)
    if split == 'train':
        print('split == train')
    return torch.utils.data.DataLoader(dataset, batch_size=batchSize, shuffle=shuffle, num_workers=int(workers))


class AverageMeter(object):
    """misc.py:121-136: plain attributes `val` (last value), `avg` (count-weighted running mean), `sum`, `count`;
    `reset()`, `update(val, n=1)`.  All four stay assignable, as in the reference."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val, self.avg, self.sum, self.count = 0, 0, 0, 0

    def update(self, val, n=1):
        self.val = val
        self.sum = self.sum + val * n
        self.count = self.count + n
        self.avg = self.sum / self.count if self.count else 0


class ImagePool:
    """misc.py:140-161 semantics (the pix2pix history buffer): the first `pool_size` queries are stored and returned
    as they are; afterwards a fair coin decides between returning the new image untouched and swapping it with a
    uniformly chosen stored one, which is returned instead.  `pool_size == 0` disables the pool.

    The history is a list with one entry per slot, like the reference's (`images`), so queries may differ in shape, dtype
    or device (a ragged last batch -- getLoader has no drop_last -- or a change of imageSize).  A slot whose stored
    tensor already has the new image's shape / dtype / device is overwritten in place (no allocation in steady state);
    any other slot is replaced by a clone.  Nothing is ever broadcast.  The draws come from an explicit numpy Generator:
    pass `seed`, or let it be drawn from numpy's global RNG so that `np.random.seed()` (which
    `datasets.pix2pix(seed=...)` calls) still makes a run reproducible."""

    def __init__(self, pool_size=50, seed=None):
        self.pool_size = pool_size
        self.num_imgs = 0
        self.images = []
        if seed is None:
            seed = int(np.random.randint(0, 2 ** 31 - 1))
        self.rng = np.random.default_rng(seed)

    def query(self, image):
        if self.pool_size == 0:
            return image
        if self.num_imgs < self.pool_size:
            self.images.append(image.detach().clone())
            self.num_imgs += 1
            return image
        if self.rng.random() <= 0.5:
            return image
        slot = int(self.rng.integers(self.pool_size))
        old = self.images[slot]
        if old.shape == image.shape and old.dtype == image.dtype and old.device == image.device:
            out = old.clone()
            old.copy_(image.detach())
        else:
            out = old
            self.images[slot] = image.detach().clone()
        return out


def adjust_learning_rate(optimizer, init_lr, epoch, factor, every):
    """misc.py:164-172: linear decay by init_lr/every per call; `epoch` and `factor` are ignored."""
    lrd = init_lr / every
    lr = max(optimizer.param_groups[0]['lr'] - lrd, 0)
    for param_group in optimizer.param_groups:
        param_group['lr'] = lr


def to_uint8_image(tensor, normalize=True):
    """CHW float tensor -> HWC uint8 with torchvision.utils.save_image(normalize=True, scale_each=False)
    semantics: clamp to [min, max], (x - min) / max(max - min, 1e-5), then *255 + 0.5, clamp, truncate
    (the rounding rule of torchvision >= 0.3; SURVEY 8c records the choice)."""
    t = tensor.detach().float().cpu().clone()
    if normalize:
        lo, hi = float(t.min()), float(t.max())
        t.clamp_(lo, hi).sub_(lo).div_(max(hi - lo, 1e-5))
    return t.mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to(torch.uint8).numpy()


def save_image(tensor, path, normalize=True, scale_each=False):
    from PIL import Image
    arr = to_uint8_image(tensor, normalize)
    Image.fromarray(arr[..., 0] if arr.shape[2] == 1 else arr).save(path)
