"""MI355X-native `loss` module: the low/high-frequency extractors of the Fusion-discriminator.

The reference ships this file only as orphaned bytecode (/root/reference/__pycache__/
loss.cpython-36.pyc; source `loss.py` absent).  Names, constructor signatures and semantics
follow the disassembly in SURVEY.md Appendix B: `Blur` (loss.py:122-151),
`isotropic_gaussian_kernel` (:153-159), module-level `blur_kernel` / `blur` (:161-162),
`get_laplacian_kernel2d` (:205-241), `Laplacian` (:245-301), `laplace_filter` (:304).
The math runs in hand-written gfx950 kernels (csrc/freqsplit.hip); inputs must be on the GPU.
The reference hard-codes `.cuda()` in these classes, so it has no CPU path either.
"""
import numpy as np
import torch
import torch.nn as nn

from fdgan_hip import engine as E


def isotropic_gaussian_kernel(l, sigma, tensor=True):
    """loss.py:153-159.  Normalised l x l Gaussian (host-side, tiny)."""
    ax = np.arange(-l // 2 + 1.0, l // 2 + 1.0)
    xx, yy = np.meshgrid(ax, ax)
    k = np.exp(-(xx ** 2 + yy ** 2) / (2.0 * sigma ** 2))
    k = k / np.sum(k)
    return torch.FloatTensor(k) if tensor else k


def get_laplacian_kernel2d(k):
    """loss.py:205-241: ones(k, k) with centre 1 - k^2 (NOT normalised)."""
    K = torch.ones(k, k)
    K[k // 2, k // 2] = 1 - k ** 2
    return K


def _gpu_f32(x, what):
    E.require_gpu(x, what)
    if x.dim() != 4:
        raise ValueError("Invalid input shape, we expect BxCxHxW. Got: {}".format(tuple(x.shape)))
    return x.detach().float().contiguous()


class Blur(nn.Module):
    """loss.py:122-151.  The HIP kernel implements the instance the reference builds (l = 15, the
    sigma = 3 isotropic Gaussian, :161-162); other kernels are rejected rather than approximated."""

    def __init__(self, l=15, kernel=None, use_input_norm=True):
        super().__init__()
        self.l = l
        self.use_input_norm = use_input_norm
        ref = isotropic_gaussian_kernel(15, 3.0)
        kernel = ref if kernel is None else torch.as_tensor(kernel, dtype=torch.float32)
        if l != 15 or tuple(kernel.shape[-2:]) != (15, 15) or (kernel.reshape(15, 15) - ref).abs().max() > 1e-7:
            raise NotImplementedError("the HIP Blur implements l=15, isotropic_gaussian_kernel(15, 3.0) (loss.py:161)")
        self.register_buffer("kernel", kernel.view(1, 1, l, l))
        if use_input_norm:
            self.register_buffer("mean", torch.Tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1))
            self.register_buffer("std", torch.Tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1))

    def forward(self, x):
        if torch.is_grad_enabled() and x.requires_grad:
            return _BlurFn.apply(x, self.use_input_norm)
        return E.blur15(_gpu_f32(x, "Blur.forward"), self.use_input_norm)


class Laplacian(nn.Module):
    """loss.py:245-301: depthwise conv2d with get_laplacian_kernel2d(k), zero padding (k-1)//2."""

    def __init__(self, kernel_size):
        super().__init__()
        if kernel_size != 3:
            raise NotImplementedError("the HIP Laplacian implements kernel_size=3 (loss.py:304)")
        self.kernel_size = kernel_size
        self._padding = (kernel_size - 1) // 2
        self.register_buffer("kernel", get_laplacian_kernel2d(kernel_size))

    def forward(self, x):
        if torch.is_grad_enabled() and x.requires_grad:
            return _LaplacianFn.apply(x)
        return E.laplacian3(_gpu_f32(x, "Laplacian.forward"))


blur_kernel = isotropic_gaussian_kernel(l=15, sigma=3.0)       # loss.py:161
blur = Blur(l=15, kernel=blur_kernel)                           # loss.py:162 (use_input_norm defaults True)
laplace_filter = Laplacian(kernel_size=3)                       # loss.py:304


class _BlurFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, use_input_norm):
        ctx.norm = use_input_norm
        return E.blur15(_gpu_f32(x, "Blur.forward"), use_input_norm)

    @staticmethod
    def backward(ctx, dy):
        return E.blur15_bwd(dy.detach().float().contiguous(), ctx.norm), None


class _LaplacianFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return E.laplacian3(_gpu_f32(x, "Laplacian.forward"))

    @staticmethod
    def backward(ctx, dy):
        return E.laplacian3(dy.detach().float().contiguous())     # self-adjoint: symmetric kernel, zero padding


def fusion_input(img, use_input_norm=True):
    """cat([img, LF(img), HF(img)], 1): what the Fusion-discriminator sees for an image
    (/root/reference/facades/network.png).  Returns NCHW fp32 (B, 9, H, W)."""
    if torch.is_grad_enabled() and img.requires_grad:      # generator's adversarial path: gradients flow through both filters
        E.require_gpu(img, "fusion_input")
        return torch.cat([img.float(), _BlurFn.apply(img, use_input_norm), _LaplacianFn.apply(img)], 1)
    x = _gpu_f32(img, "fusion_input")
    return torch.cat([x, E.blur15(x, use_input_norm), E.laplacian3(x)], 1)
