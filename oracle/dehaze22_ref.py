"""TEST INFRASTRUCTURE (see oracle/__init__.py).  CPU restatement of the pix2pix PatchGAN
discriminator `D` of /root/reference/models/dehaze22.py:114-156 (+ `blockUNet` :51-65) in stock
fp32 PyTorch-CPU ops.  Pinned by tests/golden/d22_2x64.npz, which the REAL reference produced
(oracle/make_golden.py), with the same name-keyed deterministic weights.

    layer1: Conv4x4 s2 p1 (nc->nf)                               :122
    layer2: LReLU(0.2), Conv4x4 s2 p1 (nf->2nf),  BN             :127  (blockUNet :51-65)
    layer3: LReLU,      Conv4x4 s2 p1 (2nf->4nf), BN             :133
    layer4: LReLU,      Conv4x4 s1 p1 (4nf->8nf), BN             :139-141
    layer5: LReLU,      Conv4x4 s1 p1 (8nf->1),   Sigmoid        :147-149
All convs bias-free; 256 -> 128 -> 64 -> 32 -> 31 -> 30.
"""
import torch.nn as nn

from .dehaze1113_ref import _Named


def _block_unet(cin, cout, name):
    inner = _Named(leakyrelu=nn.LeakyReLU(0.2, inplace=True),
                   conv=nn.Conv2d(cin, cout, 4, 2, 1, bias=False),
                   bn=nn.BatchNorm2d(cout))
    return _Named(**{name: inner})


class D(nn.Module):
    def __init__(self, nc, nf):
        super().__init__()
        self.main = _Named(
            layer1=_Named(conv=nn.Conv2d(nc, nf, 4, 2, 1, bias=False)),
            layer2=_block_unet(nf, nf * 2, "layer2"),
            layer3=_block_unet(nf * 2, nf * 4, "layer3"),
            layer4=_Named(leakyrelu=nn.LeakyReLU(0.2, inplace=True),
                          conv=nn.Conv2d(nf * 4, nf * 8, 4, 1, 1, bias=False),
                          bn=nn.BatchNorm2d(nf * 8)),
            layer5=_Named(leakyrelu=nn.LeakyReLU(0.2, inplace=True),
                          conv=nn.Conv2d(nf * 8, 1, 4, 1, 1, bias=False),
                          sigmoid=nn.Sigmoid()),
        )

    def forward(self, x):
        return self.main(x)
