"""TEST INFRASTRUCTURE (oracle).  DenseNet-121 `features` topology, CPU fp32.

The reference pulls its encoder out of torchvision's DenseNet-121
(/root/reference/models/dehaze1113.py:707-728: `haze_class.features.conv0`,
`.relu0`, `.denseblock1-4`, `.transition1-3`, `.norm5`).  torchvision is a
third-party dependency that is neither vendored in the reference tree nor
installed in this image (pinned only as "conda install pytorch=0.3.0
torchvision", /root/reference/README.md:24), so its published architecture is
restated here: growth 32, bn_size 4, block config (6, 12, 24, 16), 64 initial
features; every conv bias-free; sub-module names `norm1 relu1 conv1 norm2 relu2
conv2` per dense layer and `norm relu conv pool` per transition -- the names
the reference's state_dict keys carry.

Parity status: topology unpinned by the reference itself (no test, no vector);
pinned by key names and parameter counts (tests/test_oracle_golden.py).
"""
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F

GROWTH = 32
BN_SIZE = 4
BLOCK_CONFIG = (6, 12, 24, 16)
INIT_FEATURES = 64


class DenseLayer(nn.Module):
    """BN -> ReLU -> 1x1 (Cin -> 4*growth) -> BN -> ReLU -> 3x3 (-> growth)."""

    def __init__(self, cin):
        super().__init__()
        self.norm1 = nn.BatchNorm2d(cin)
        self.relu1 = nn.ReLU(inplace=True)
        self.conv1 = nn.Conv2d(cin, BN_SIZE * GROWTH, 1, 1, 0, bias=False)
        self.norm2 = nn.BatchNorm2d(BN_SIZE * GROWTH)
        self.relu2 = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(BN_SIZE * GROWTH, GROWTH, 3, 1, 1, bias=False)

    def forward(self, feats):
        x = torch.cat(feats, 1) if isinstance(feats, (list, tuple)) else feats
        y = self.conv1(self.relu1(self.norm1(x)))
        return self.conv2(self.relu2(self.norm2(y)))


class DenseBlock(nn.ModuleDict):
    def __init__(self, nlayers, cin):
        super().__init__()
        for i in range(nlayers):
            self["denselayer%d" % (i + 1)] = DenseLayer(cin + i * GROWTH)

    def forward(self, x):
        feats = [x]
        for layer in self.values():
            feats.append(layer(feats))
        return torch.cat(feats, 1)


class Transition(nn.Sequential):
    def __init__(self, cin, cout):
        super().__init__(OrderedDict([
            ("norm", nn.BatchNorm2d(cin)),
            ("relu", nn.ReLU(inplace=True)),
            ("conv", nn.Conv2d(cin, cout, 1, 1, 0, bias=False)),
            ("pool", nn.AvgPool2d(2, 2)),
        ]))


class DenseNet121(nn.Module):
    """Only `.features` is used by the reference (dehaze1113.py:709-728)."""

    def __init__(self):
        super().__init__()
        mods = OrderedDict()
        mods["conv0"] = nn.Conv2d(3, INIT_FEATURES, 7, 2, 3, bias=False)
        mods["norm0"] = nn.BatchNorm2d(INIT_FEATURES)
        mods["relu0"] = nn.ReLU(inplace=True)
        mods["pool0"] = nn.MaxPool2d(3, 2, 1)
        c = INIT_FEATURES
        for b, n in enumerate(BLOCK_CONFIG):
            mods["denseblock%d" % (b + 1)] = DenseBlock(n, c)
            c += n * GROWTH
            if b != len(BLOCK_CONFIG) - 1:
                mods["transition%d" % (b + 1)] = Transition(c, c // 2)
                c //= 2
        mods["norm5"] = nn.BatchNorm2d(c)
        self.features = nn.Sequential(mods)
        self.classifier = nn.Linear(c, 1000)

    def forward(self, x):
        f = F.relu(self.features(x), inplace=True)
        f = F.adaptive_avg_pool2d(f, (1, 1)).flatten(1)
        return self.classifier(f)


def densenet121(pretrained=False, **_):
    """Signature of torchvision.models.densenet121; `pretrained` is ignored
    (no network, no weights in the tree -- SURVEY section 0)."""
    return DenseNet121()
