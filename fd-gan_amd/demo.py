"""Inference driver with the reference's command line (/root/reference/demo.py):

    python demo.py --valDataroot ./testsample1 --netG ./test_model/netG_epoch_XX.pth

Loads the generator checkpoint (keys saved under nn.DataParallel carry a `module.` prefix, :78-86),
runs `models.dehaze1113.FDGAN` -- in TRAIN mode, as the reference does (README.md:38; no .eval()) --
over the `pix2pix` dataset and writes ./result_AAAI20/image/<running index>.png with min-max
normalisation (:141-151).  The generator runs on the MI355X through libfdgan_hip.so; there is no
CPU path.  Training-only flags are accepted for command-line compatibility.
"""
from __future__ import print_function

import argparse
import os
import time
from collections import OrderedDict

import torch

import models.dehaze1113 as net
from misc import getLoader, save_image


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument('--dataset', required=False, default='pix2pix', help='')
    p.add_argument('--valDataroot', required=False, default="", help='path to val dataset')
    p.add_argument('--mode', type=str, default='B2A', help='B2A: facade, A2B: edges2shoes')
    p.add_argument('--valBatchSize', type=int, default=1, help='input batch size')
    p.add_argument('--batchSize', type=int, default=1, help='input batch size')
    p.add_argument('--originalSize', type=int, default=1024, help='the height / width of the original input image')
    p.add_argument('--imageSize', type=int, default=1024, help='the height / width of the cropped input image to network')
    p.add_argument('--inputChannelSize', type=int, default=3, help='size of the input channels')
    p.add_argument('--outputChannelSize', type=int, default=3, help='size of the output channels')
    p.add_argument('--lrD', type=float, default=0.0002, help='learning rate, default=0.0002')
    p.add_argument('--lrG', type=float, default=0.0002, help='learning rate, default=0.0002')
    p.add_argument('--netG', default='', help="path to netG (to continue training)")
    p.add_argument('--beta1', type=float, default=0.5, help='beta1 for adam')
    p.add_argument('--netD', default='', help="path to netD (to continue training)")
    p.add_argument('--workers', type=int, help='number of data loading workers', default=1)
    p.add_argument('--display', type=int, default=5, help='interval for displaying train-logs')
    p.add_argument('--evalIter', type=int, default=500, help='interval for evauating(generating) images from valDataroot')
    p.add_argument('--outDir', default='./result_AAAI20/image/', help='(addition) where the PNGs go')
    return p


def load_generator_state(path):
    """demo.py:78-86 strips exactly 7 characters (`module.`) from every key; checkpoints saved without
    DataParallel are accepted as they are.  torchvision-0.2 era spellings of the dense-layer keys
    (`norm.1`, `conv.2`, ...) are mapped to the current ones and missing `num_batches_tracked`
    buffers are tolerated."""
    import re
    state_dict = torch.load(path, map_location='cpu')
    out = OrderedDict()
    for k, v in state_dict.items():
        name = k[7:] if k.startswith('module.') else k
        name = re.sub(r'\.(norm|relu|conv)\.([12])\.', r'.\1\2.', name)
        out[name] = v
    return out


def run(opt):
    from fdgan_hip.dp import DpContext
    dp = DpContext.from_env()            # torchrun: one process per GPU, every world-th image each (demo.py:89's
    dev = dp.device or torch.device('cuda', 0)   # nn.DataParallel split of the batch, without the gather)
    loader = getLoader(opt.dataset, opt.valDataroot, opt.imageSize, opt.imageSize, opt.valBatchSize, opt.workers,
                       mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5), split='Train', shuffle=False, seed=None)
    names = None
    if dp.world > 1:
        if opt.valBatchSize != 1:
            raise ValueError("multi-GPU demo shards single images: use --valBatchSize 1")
        names = dp.item_indices(len(loader.dataset))
        loader = torch.utils.data.DataLoader(torch.utils.data.Subset(loader.dataset, names), batch_size=1,
                                             shuffle=False, num_workers=int(opt.workers))
    import warnings
    with warnings.catch_warnings():           # "encoder randomly initialised": the checkpoint below overwrites every tensor
        warnings.simplefilter("ignore")
        netG = net.FDGAN()
    sd = load_generator_state(opt.netG)
    missing = [k for k in netG.state_dict() if k not in sd]
    if any(not k.endswith('num_batches_tracked') for k in missing):
        raise KeyError("checkpoint lacks keys: %s" % [k for k in missing if not k.endswith('num_batches_tracked')][:5])
    netG.load_state_dict(sd, strict=False)
    netG = netG.to(dev)          # stays in train mode on purpose (README.md:38)
    os.makedirs(opt.outDir, exist_ok=True)
    index = -1
    written = []
    for i, data in enumerate(loader, 0):
        input_cpu, target_cpu = data
        inp = input_cpu.float().to(dev)
        torch.cuda.synchronize()
        start = time.time()
        with torch.no_grad():
            x_hat = netG(inp)
        torch.cuda.synchronize()     # the reference times without a device sync (:131-135)
        print(time.time() - start)
        for _ in range(opt.valBatchSize):
            index += 1
            gidx = index if names is None else names[index]
            print(gidx)
            path = os.path.join(opt.outDir, str(gidx) + '.png')
            save_image(x_hat[0], path, normalize=True, scale_each=False)   # always element 0 (:141,148)
            written.append(path)
    dp.close()
    return written


if __name__ == '__main__':
    opt = build_parser().parse_args()
    print(opt)
    run(opt)
