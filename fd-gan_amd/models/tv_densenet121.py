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
    """`pretrained=True` (what /root/reference/models/dehaze1113.py:707 asks torchvision for) cannot be honoured offline: the
    encoder comes back RANDOMLY initialised and a warning says so -- load the authors' checkpoint (demo.py --netG) or a
    torchvision densenet121 state_dict with `load_densenet121_weights` before training on the result."""
    if pretrained:
        import warnings
        warnings.warn("densenet121(pretrained=True): ImageNet weights are not available offline; the DenseNet-121 encoder is "
                      "randomly initialised.  Load a checkpoint (FDGAN.load_state_dict / models.tv_densenet121."
                      "load_densenet121_weights) before relying on it.", stacklevel=2)
    return DenseNet121()


def load_densenet121_weights(fdgan, path):
    """Copies a torchvision densenet121 state_dict (features.conv0 / denseblockN / transitionN / norm5; both the 0.2-era
    `norm.1` and the current `norm1` spellings) into the encoder modules FDGAN pulled out of it
    (dehaze1113.py:707-728: conv0, dense_block1-3, trans_block1-3 as used by forward)."""
    import re
    import torch
    sd = torch.load(path, map_location="cpu")
    sd = sd.get("state_dict", sd) if isinstance(sd, dict) else sd
    ren = {"features.conv0.": "conv0.", "features.norm0.": "norm0.", "features.denseblock1.": "dense_block1.",
           "features.denseblock2.": "dense_block2.", "features.denseblock3.": "dense_block3.",
           "features.transition1.": "trans_block1.", "features.transition2.": "trans_block2.", "features.transition3.": "trans_block3."}
    own = fdgan.state_dict()
    picked = {}
    for k, v in sd.items():
        k = re.sub(r"\.(norm|relu|conv)\.([12])\.", r".\1\2.", k[7:] if k.startswith("module.") else k)
        for src, dst in ren.items():
            if k.startswith(src) and dst + k[len(src):] in own:
                picked[dst + k[len(src):]] = v
    if not picked:
        raise KeyError("%s holds no torchvision densenet121 encoder keys (features.denseblock1...)" % path)
    missing = [k for k in own if k.split(".")[0] in ("dense_block1", "dense_block2", "dense_block3", "trans_block1", "trans_block2",
                                                     "trans_block3") and k not in picked and not k.endswith("num_batches_tracked")]
    if missing:
        raise KeyError("densenet121 checkpoint lacks %d encoder tensors, e.g. %s" % (len(missing), missing[:3]))
    fdgan.load_state_dict(picked, strict=False)
    fdgan.encoder_weights_loaded = True
    return fdgan
