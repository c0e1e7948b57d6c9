"""TEST INFRASTRUCTURE (oracle).  CPU fp32 restatement of
/root/reference/myutils/vgg16.py:6-49 (perceptual-loss feature extractor)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

# (name, cin, cout); "P" = 2x2 max-pool; "T" = tap the current activation.
_PLAN = [("conv1_1", 3, 64), ("conv1_2", 64, 64), "T", "P",
         ("conv2_1", 64, 128), ("conv2_2", 128, 128), "T", "P",
         ("conv3_1", 128, 256), ("conv3_2", 256, 256), ("conv3_3", 256, 256), "T", "P",
         ("conv4_1", 256, 512), ("conv4_2", 512, 512), ("conv4_3", 512, 512), "T"]
_UNUSED = [("conv5_1", 512, 512), ("conv5_2", 512, 512), ("conv5_3", 512, 512)]   # vgg16.py:23-25


class Vgg16(nn.Module):
    def __init__(self):
        super().__init__()
        for item in _PLAN + _UNUSED:
            if isinstance(item, tuple):
                setattr(self, item[0], nn.Conv2d(item[1], item[2], 3, 1, 1))

    def forward(self, X):
        """Returns [relu1_2, relu2_2, relu3_3, relu4_3] (vgg16.py:27-49)."""
        h, outs = X, []
        for item in _PLAN:
            if item == "T":
                outs.append(h)
            elif item == "P":
                h = F.max_pool2d(h, 2, 2)
            else:
                h = F.relu(getattr(self, item[0])(h))
        return outs
