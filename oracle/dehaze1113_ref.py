"""TEST INFRASTRUCTURE (oracle).  CPU fp32 restatement of the reference's
`models/dehaze1113.py` hot path: generator `FDGAN`, Fusion-discriminator `D`,
and the decoder blocks, with state_dict keys identical to the reference's
 so weights can be exchanged by `load_state_dict`.

Each function cites the reference lines it follows.  Validated against the
imported reference by oracle/make_golden.py (max |diff| recorded in
tests/golden/MANIFEST.json).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .densenet121 import densenet121


class BottleneckBlockdy(nn.Module):
    """/root/reference/models/dehaze1113.py:256-275.

    out = conv1(relu(x)); out = conv2(relu(out)); return cat([x, out]) where
    the ReLU is *in place* (:261), so the concatenated `x` is relu(x) and the
    caller's tensor is overwritten.  bn1/bn2 are registered (:260,:264) but
    never called.
    """

    def __init__(self, in_planes, out_planes, dropRate=0.0):
        super().__init__()
        inter = out_planes * 4
        self.bn1 = nn.BatchNorm2d(in_planes)
        self.conv1 = nn.Conv2d(in_planes, inter, 1, 1, 0, bias=False)
        self.bn2 = nn.BatchNorm2d(inter)
        self.conv2 = nn.Conv2d(inter, out_planes, 3, 1, 1, bias=False)
        self.droprate = dropRate        # every FDGAN call site passes the default 0 (:731-739); > 0: F.dropout after each conv (:270-274)
        self.masks = []                 # the 0 / 1 masks of the last training forward, in order (tests hand them to the HIP path)

    def _dropout(self, out):
        """F.dropout(out, p, inplace=False, training=self.training) spelled out (ATen: noise = empty_like(out).bernoulli_(1 - p);
        out * noise / (1 - p)) so that the mask it drew can be read back."""
        if self.droprate > 0 and self.training:
            m = torch.empty_like(out).bernoulli_(1.0 - self.droprate)
            self.masks.append(m)
            return out * (m / (1.0 - self.droprate))
        return out

    def forward(self, x):
        self.masks = []
        x.relu_()                       # aliasing kept on purpose (Appendix E.2)
        mid = self._dropout(self.conv1(x))
        out = self._dropout(self.conv2(torch.relu(mid)))
        return torch.cat([x, out], 1)


class TransitionBlockdy(nn.Module):
    """/root/reference/models/dehaze1113.py:358-370: relu (in place) ->
    ConvTranspose2d 1x1 (weight (Cin,Cout,1,1), no bias) -> nearest x2."""

    def __init__(self, in_planes, out_planes, dropRate=0.0):
        super().__init__()
        self.bn1 = nn.BatchNorm2d(in_planes)    # registered, unused (:361)
        self.conv1 = nn.ConvTranspose2d(in_planes, out_planes, 1, 1, 0, bias=False)
        self.droprate = dropRate        # > 0: F.dropout between the conv and the upsample (:367-368)
        self.masks = []

    def forward(self, x):
        self.masks = []
        y = self.conv1(x.relu_())
        if self.droprate > 0 and self.training:
            m = torch.empty_like(y).bernoulli_(1.0 - self.droprate)
            self.masks.append(m)
            y = y * (m / (1.0 - self.droprate))
        return F.interpolate(y, scale_factor=2, mode="nearest")   # F.upsample_nearest (:370)


class FDGAN(nn.Module):
    """/root/reference/models/dehaze1113.py:702-801."""

    def __init__(self):
        super().__init__()
        feats = densenet121(pretrained=True).features           # :707
        self.conv0 = feats.conv0                                 # :709 (never called)
        self.relu0 = feats.relu0                                 # :710
        self.dense_block1 = feats.denseblock1                    # :713
        self.trans_block1 = feats.transition1
        self.dense_block2 = feats.denseblock2                    # :717
        self.trans_block2 = feats.transition2
        self.dense_block3 = feats.denseblock3                    # :721
        self.trans_block3 = feats.transition3
        self.dense_block31 = feats.denseblock4                   # :725 (never called)
        self.dense_norm31 = feats.norm5                          # :728 (never called)
        self.dense_block4 = BottleneckBlockdy(512, 256)          # :731
        self.trans_block4 = TransitionBlockdy(768, 128)
        self.dense_block5 = BottleneckBlockdy(384, 128)          # :735
        self.trans_block5 = TransitionBlockdy(512, 64)
        self.dense_block6 = BottleneckBlockdy(64, 32)            # :739
        self.trans_block6 = TransitionBlockdy(96, 16)
        self.conv_refin1 = nn.Conv2d(3, 64, 3, 1, 1)             # :744
        self.conv_refin6 = nn.Conv2d(640, 512, 3, 1, 1)          # :746
        self.conv_refin5 = nn.Conv2d(256, 128, 1, 1, 0)          # :747
        self.tanh = nn.Tanh()
        self.conv_refin3 = nn.Conv2d(16, 3, 3, 1, 1)             # :749
        self.conv_refin2 = nn.Conv2d(64, 32, 1, 1, 0)            # :751
        self.conv_refine4 = nn.Conv2d(160, 128, 3, 1, 1)         # :755

    def forward(self, x, taps=None):
        """`taps`, if a dict, receives named intermediates (for golden files)."""
        def tap(name, t):
            if taps is not None:
                taps[name] = t.detach().clone()
            return t
        x0 = tap("x0", self.relu0(self.conv_refin1(x)))                       # :760
        x01 = tap("x01", self.conv_refin2(F.avg_pool2d(x0, 2)))               # :763
        x1 = tap("x1", self.trans_block1(self.dense_block1(x0)))              # :767-769
        x10 = tap("x10", self.conv_refine4(torch.cat([x01, x1], 1)))          # :773
        x2 = tap("x2", self.trans_block2(self.dense_block2(x10)))             # :774
        x3 = tap("x3", self.trans_block3(self.dense_block3(x2)))              # :778
        x22 = tap("x22", self.conv_refin5(F.avg_pool2d(x2, 2)))               # :780
        x4 = self.trans_block4(self.dense_block4(
            self.conv_refin6(torch.cat([x3, x22], 1))))                       # :783
        tap("x4", x4)
        x5 = tap("x5", self.trans_block5(self.dense_block5(torch.cat([x4, x2], 1))))   # :786-790
        x6 = tap("x6", self.trans_block6(self.dense_block6(x5)))              # :795
        return self.tanh(self.conv_refin3(x6))                                # :799


class _Named(nn.Sequential):
    """Sequential whose children are added by name (keeps reference key names
    such as `main.layer2.layer2.conv.weight` without dotted child names, which
    modern torch rejects -- SURVEY Appendix D)."""

    def __init__(self, **children):
        super().__init__()
        for k, v in children.items():
            self.add_module(k, v)


def _block_unet1(cin, cout, name):
    """/root/reference/models/dehaze1113.py:29-43 with transposed=False,
    bn=True, relu=False, dropout=False (the only form D uses, :201,:207):
    LeakyReLU(0.2) -> Conv3x3 s1 p1 (no bias) -> BatchNorm."""
    inner = _Named(leakyrelu=nn.LeakyReLU(0.2, inplace=True),
                   conv=nn.Conv2d(cin, cout, 3, 1, 1, bias=False),
                   bn=nn.BatchNorm2d(cout))
    return _Named(**{name: inner})


class D(nn.Module):
    """Fusion-discriminator, /root/reference/models/dehaze1113.py:188-230.

    Conv4x4 s2 (nc->nf) | LReLU, Conv3x3 (nf->2nf), BN | LReLU, Conv3x3
    (2nf->4nf), BN | LReLU, Conv4x4 s1 (4nf->8nf) [BN commented out :215] |
    LReLU, Conv4x4 s1 (8nf->1), Sigmoid.  All convs bias-free.
    """

    def __init__(self, nc, nf):
        super().__init__()
        self.main = _Named(
            layer1=_Named(conv=nn.Conv2d(nc, nf, 4, 2, 1, bias=False)),          # :196
            layer2=_block_unet1(nf, nf * 2, "layer2"),                             # :201
            layer3=_block_unet1(nf * 2, nf * 4, "layer3"),                         # :207
            layer4=_Named(leakyrelu=nn.LeakyReLU(0.2, inplace=True),               # :213
                          conv=nn.Conv2d(nf * 4, nf * 8, 4, 1, 1, bias=False)),    # :214
            layer5=_Named(leakyrelu=nn.LeakyReLU(0.2, inplace=True),               # :221
                          conv=nn.Conv2d(nf * 8, 1, 4, 1, 1, bias=False),          # :222
                          sigmoid=nn.Sigmoid()),                                   # :223
        )

    def forward(self, x):
        return self.main(x)
