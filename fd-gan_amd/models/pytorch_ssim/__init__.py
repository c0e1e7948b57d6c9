"""`models.pytorch_ssim` surface of the reference (/root/reference/models/pytorch_ssim/__init__.py): `ssim(img1, img2)`
and the `SSIM` module, on the HIP path (csrc/ssim.hip).  Differentiable w.r.t. `img1` (the generated image); the
reference's window (sigma 1.5) in any odd size <= 11 (default 11); size_average True (a scalar) and False (one value per image).
"""
import ctypes as C

import torch

from fdgan_hip import engine as E
from fdgan_hip import lib as L


def _check_window(window_size):
    if not (isinstance(window_size, int) and 1 <= window_size <= 11 and window_size % 2 == 1):
        raise NotImplementedError("the HIP SSIM takes odd window sizes <= 11 (the reference's default is 11), got %r" % (window_size,))


def _run_fwd(x, y, size_average=True, window_size=11):
    n, c, h, w = x.shape
    tiles = ((h + 31) // 32) * ((w + 31) // 32)
    partial = torch.empty(n * c * tiles, dtype=torch.float32, device=x.device)
    d = [torch.empty_like(x) for _ in range(3)]
    L.check(L.load().fdgan_ssim_fwd_w(x.data_ptr(), y.data_ptr(), n * c, h, w, window_size, partial.data_ptr(), partial.numel(),
                                      d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), E.stream_ptr()), "ssim_fwd")
    if size_average:
        return partial.double().sum().float() / (n * c * h * w), d
    # size_average=False (:36-37): ssim_map.mean(1).mean(1).mean(1) -- one value per image; the kernel's partial sums are
    # per (plane, tile), planes in (image, channel) order
    return partial.view(n, c * tiles).double().sum(1).float() / (c * h * w), d


class _SsimFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img1, img2, size_average, window_size=11):
        x, y = img1.detach().float().contiguous(), img2.detach().float().contiguous()
        val, d = _run_fwd(x, y, size_average, window_size)
        ctx.save_for_backward(x, y, *d)
        ctx.size_average, ctx.window_size = size_average, window_size
        return val

    @staticmethod
    def backward(ctx, g):
        x, y, da, db, dc = ctx.saved_tensors
        n, c, h, w = x.shape
        dx = torch.empty_like(x)
        weight = 1.0 / (n * c * h * w) if ctx.size_average else 1.0 / (c * h * w)
        L.check(L.load().fdgan_ssim_bwd_w(x.data_ptr(), y.data_ptr(), da.data_ptr(), db.data_ptr(), dc.data_ptr(), n * c, h, w,
                                          ctx.window_size, C.c_float(weight), dx.data_ptr(), E.stream_ptr()), "ssim_bwd")
        # the upstream gradient (a scalar, or one value per image) stays on the device: no host sync inside backward()
        return dx.mul_(g if ctx.size_average else g.view(n, 1, 1, 1)), None, None, None


def ssim(img1, img2, window_size=11, size_average=True):
    """pytorch_ssim/__init__.py:65-73."""
    E.require_gpu(img1, "ssim")
    E.require_gpu(img2, "ssim")
    _check_window(window_size)
    if img1.shape != img2.shape or img1.dim() != 4:
        raise ValueError("ssim expects two NCHW tensors of the same shape")
    if img2.requires_grad and torch.is_grad_enabled():
        raise NotImplementedError("ssim is differentiable w.r.t. its first argument (the generated image) only")
    if torch.is_grad_enabled() and img1.requires_grad:
        return _SsimFn.apply(img1, img2, bool(size_average), window_size)
    return _run_fwd(img1.detach().float().contiguous(), img2.detach().float().contiguous(), bool(size_average), window_size)[0]


class SSIM(torch.nn.Module):
    """pytorch_ssim/__init__.py:39-63."""

    def __init__(self, window_size=11, size_average=True):
        super().__init__()
        self.window_size, self.size_average = window_size, size_average

    def forward(self, img1, img2):
        return ssim(img1, img2, self.window_size, self.size_average)
