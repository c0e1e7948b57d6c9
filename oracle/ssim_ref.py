"""TEST INFRASTRUCTURE (oracle).  CPU fp32 restatement of the differentiable
SSIM loss, /root/reference/models/pytorch_ssim/__init__.py:8-73 (11x11 Gaussian
sigma 1.5 window, depthwise conv, zero padding 5, C1=0.01^2, C2=0.03^2)."""
from math import exp

import torch
import torch.nn.functional as F


def gaussian_window(size=11, sigma=1.5):
    """pytorch_ssim/__init__.py:8-10."""
    g = torch.tensor([exp(-(i - size // 2) ** 2 / float(2 * sigma ** 2)) for i in range(size)])
    return g / g.sum()


def ssim(img1, img2, window_size=11, size_average=True):
    """pytorch_ssim/__init__.py:17-37 and :65-73."""
    C = img1.shape[1]
    g = gaussian_window(window_size).unsqueeze(1)
    win = (g @ g.t()).float().expand(C, 1, window_size, window_size).contiguous().to(img1.dtype)
    pad = window_size // 2

    def blur(t):
        return F.conv2d(t, win, padding=pad, groups=C)
    mu1, mu2 = blur(img1), blur(img2)
    mu1_sq, mu2_sq, mu12 = mu1 * mu1, mu2 * mu2, mu1 * mu2
    s1 = blur(img1 * img1) - mu1_sq
    s2 = blur(img2 * img2) - mu2_sq
    s12 = blur(img1 * img2) - mu12
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    m = ((2 * mu12 + C1) * (2 * s12 + C2)) / ((mu1_sq + mu2_sq + C1) * (s1 + s2 + C2))
    return m.mean() if size_average else m.mean(1).mean(1).mean(1)
