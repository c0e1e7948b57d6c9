"""Run helpers behind the reference's `misc` module (/root/reference/misc.py): loader factory and
the GAN-training utilities.  Semantics per SURVEY Appendix G; image saving follows
torchvision.utils.save_image(normalize=True) (demo.py:151), which the reference imports from
torchvision (absent here).
"""
import os

import numpy as np
import torch


def create_exp_dir(exp):
    try:
        os.makedirs(exp)
        print('Creating exp dir: %s' % exp)
    except OSError:
        pass
    return True


def weights_init(m):
    """misc.py:16-22: class-name substring dispatch ('Conv' also matches ConvTranspose2d)."""
    classname = m.__class__.__name__
    if classname.find('Conv') != -1:
        m.weight.data.normal_(0.0, 0.02)
    elif classname.find('BatchNorm') != -1:
        m.weight.data.normal_(1.0, 0.02)
        m.bias.data.fill_(0)


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
    """misc.py:121-136."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


class ImagePool:
    """misc.py:140-161: history of generated images (pix2pix); uses numpy's global RNG like the reference."""

    def __init__(self, pool_size=50):
        self.pool_size = pool_size
        if pool_size > 0:
            self.num_imgs = 0
            self.images = []

    def query(self, image):
        if self.pool_size == 0:
            return image
        if self.num_imgs < self.pool_size:
            self.images.append(image.clone())
            self.num_imgs += 1
            return image
        if np.random.uniform(0, 1) > 0.5:
            random_id = np.random.randint(self.pool_size, size=1)[0]
            tmp = self.images[random_id].clone()
            self.images[random_id] = image.clone()
            return tmp
        return image


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
