"""Parameter containers with torchvision DenseNet-121's module tree.

The reference takes its encoder from `torchvision.models.densenet121(pretrained=True)`
(/root/reference/models/dehaze1113.py:707-728).  Here the same module names
(`denseblockN.denselayerM.{norm1,conv1,norm2,conv2}`, `transitionN.{norm,conv}`,
`conv0`, `norm5`) are provided so that `state_dict()` keys and shapes are identical
(SURVEY Appendix D).  These modules only OWN parameters and buffers: their math runs
in the fused HIP plan built by `models.dehaze1113.FDGAN`; calling them directly raises.
"""
from collections import OrderedDict

import torch.nn as nn

GROWTH, BN_SIZE, BLOCKS, INIT = 32, 4, (6, 12, 24, 16), 64


class _NoForward:
    def forward(self, *a, **k):
        raise RuntimeError("%s holds parameters only; its math runs inside the fused HIP plan of the "
                           "enclosing network (no eager / CPU path)" % type(self).__name__)


class _DenseLayer(_NoForward, nn.Module):
    def __init__(self, cin):
        super().__init__()
        self.norm1 = nn.BatchNorm2d(cin)
        self.relu1 = nn.ReLU(inplace=True)
        self.conv1 = nn.Conv2d(cin, BN_SIZE * GROWTH, 1, 1, 0, bias=False)
        self.norm2 = nn.BatchNorm2d(BN_SIZE * GROWTH)
        self.relu2 = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(BN_SIZE * GROWTH, GROWTH, 3, 1, 1, bias=False)


class _DenseBlock(_NoForward, nn.ModuleDict):
    def __init__(self, nlayers, cin):
        super().__init__()
        self.cin = cin
        for i in range(nlayers):
            self["denselayer%d" % (i + 1)] = _DenseLayer(cin + i * GROWTH)

    @property
    def cout(self):
        return self.cin + len(self) * GROWTH


class _Transition(_NoForward, nn.Sequential):
    def __init__(self, cin, cout):
        super().__init__(OrderedDict([("norm", nn.BatchNorm2d(cin)), ("relu", nn.ReLU(inplace=True)),
                                      ("conv", nn.Conv2d(cin, cout, 1, 1, 0, bias=False)),
                                      ("pool", nn.AvgPool2d(2, 2))]))


class _Features(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv0 = nn.Conv2d(3, INIT, 7, 2, 3, bias=False)
        self.norm0 = nn.BatchNorm2d(INIT)
        self.relu0 = nn.ReLU(inplace=True)
        self.pool0 = nn.MaxPool2d(3, 2, 1)
        c = INIT
        for b, n in enumerate(BLOCKS):
            blk = _DenseBlock(n, c)
            setattr(self, "denseblock%d" % (b + 1), blk)
            c = blk.cout
            if b + 1 < len(BLOCKS):
                setattr(self, "transition%d" % (b + 1), _Transition(c, c // 2))
                c //= 2
        self.norm5 = nn.BatchNorm2d(c)


class DenseNet121(nn.Module):
    def __init__(self):
        super().__init__()
        self.features = _Features()


def densenet121(pretrained=False, **_):
    """`pretrained` is accepted for signature compatibility; ImageNet weights are not
    available offline -- load a checkpoint with `load_state_dict` instead."""
    return DenseNet121()
