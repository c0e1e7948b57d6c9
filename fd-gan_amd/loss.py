"""MI355X-native `loss` module: the low/high-frequency extractors of the Fusion-discriminator and ContextualLoss.

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


class _CxRowsFn(torch.autograd.Function):
    """m[b, i] = max_j CX_ij of the cosine-distance matrix d (B, N, N): fdgan_cx_rows_fwd / _bwd (csrc/losses.hip)."""

    @staticmethod
    def forward(ctx, d, sigma, eps):
        import ctypes as C
        from fdgan_hip import lib as L
        dd = d.detach().float().contiguous()
        B, N, M = dd.shape
        m = torch.empty((B, N), dtype=torch.float32, device=dd.device)
        dmin, S = torch.empty_like(m), torch.empty_like(m)
        jmin = torch.empty((B, N), dtype=torch.int32, device=dd.device)
        L.check(L.load().fdgan_cx_rows_fwd(dd.data_ptr(), B * N, M, sigma, eps, m.data_ptr(), dmin.data_ptr(), S.data_ptr(),
                                           jmin.data_ptr(), E.stream_ptr()), "cx_rows_fwd")
        ctx.save_for_backward(dd, dmin, S, jmin)
        ctx.sigma, ctx.eps = sigma, eps
        return m

    @staticmethod
    def backward(ctx, gm):
        from fdgan_hip import lib as L
        dd, dmin, S, jmin = ctx.saved_tensors
        B, N, M = dd.shape
        gd = torch.empty_like(dd)
        g = gm.detach().float().contiguous()
        L.check(L.load().fdgan_cx_rows_bwd(dd.data_ptr(), B * N, M, ctx.sigma, ctx.eps, dmin.data_ptr(), S.data_ptr(), jmin.data_ptr(),
                                           g.data_ptr(), gd.data_ptr(), E.stream_ptr()), "cx_rows_bwd")
        return gd, None, None


class ContextualLoss(nn.Module):
    """loss.py:23-73 of the reference (bytecode only): same constructor, same methods.

    `forward(I, T)` = CX(cos_similarity(I, T)) on (B, C, H, W) feature maps.  The centring / L2 normalisation and the final
    -log / means are a handful of elementwise device ops; the cosine matrix is one batched library GEMM (B x HW x HW x C,
    fp32); everything between the matrix and the per-row maximum -- relative_distances, weighted_average_distances, the max
    over j: three HW x HW intermediates in the reference -- is ONE fused HIP pass per row, forward and backward
    (fdgan_cx_rows_*).  GPU tensors only: the reference hard-codes .cuda() throughout this module as well."""

    def __init__(self, sigma=0.1, b=1.0, epsilon=1e-5, similarity='cos'):
        super().__init__()
        self.sigma, self.similarity, self.b, self.e = sigma, similarity, b, epsilon

    def cos_similarity(self, image_features, target_features):
        E.require_gpu(image_features, "ContextualLoss")
        E.require_gpu(target_features, "ContextualLoss")
        if image_features.shape != target_features.shape or image_features.dim() != 4:
            raise ValueError("ContextualLoss expects two (B, C, H, W) tensors of the same shape")
        B, C = image_features.size(0), image_features.size(1)
        i = image_features.float().reshape(B, C, -1).permute(0, 2, 1)
        t = target_features.float().reshape(B, C, -1).permute(0, 2, 1)
        mu = t.mean(dim=1, keepdim=True)
        ic, tc = i - mu, t - mu
        il = ic / torch.sqrt((ic * ic).sum(dim=2, keepdim=True))
        tl = tc / torch.sqrt((tc * tc).sum(dim=2, keepdim=True))
        return 1 - torch.bmm(il, tl.permute(0, 2, 1))

    def L2_similarity(self, image_features, target_features):      # a stub in the reference too (loss.py:46-47)
        pass

    def relative_distances(self, distances):
        return distances / (distances.min(dim=2, keepdim=True)[0] + self.e)

    def weighted_average_distances(self, distances_normalized):
        w = torch.exp((self.b - distances_normalized) / self.sigma)
        return w / w.sum(dim=2, keepdim=True)

    def CX(self, distances):
        E.require_gpu(distances, "ContextualLoss.CX")
        m = _CxRowsFn.apply(distances, float(self.sigma), float(self.e))     # == max_j weighted_average(relative(d)); b cancels
        return torch.mean(-torch.log(torch.mean(m, dim=1)))

    def forward(self, image_features, target_features):
        if self.similarity != 'cos':
            raise NotImplementedError("only similarity='cos' exists in the reference (L2_similarity is `pass`)")
        return self.CX(self.cos_similarity(image_features, target_features))


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
    """loss.py:122-151.  The HIP kernel is separable and 15 taps wide: it takes what `isotropic_gaussian_kernel(l, sigma)` builds for
    any odd l <= 15 and any sigma (the reference's instance is l = 15, sigma = 3, :161-162); a kernel that is not such a Gaussian is
    rejected rather than approximated."""

    def __init__(self, l=15, kernel=None, use_input_norm=True):
        super().__init__()
        self.l = l
        self.use_input_norm = use_input_norm
        if kernel is None:
            kernel = isotropic_gaussian_kernel(l, 3.0)
        kernel = torch.as_tensor(kernel, dtype=torch.float32)
        if l < 1 or l > 15 or l % 2 == 0 or tuple(kernel.shape[-2:]) != (l, l):
            raise NotImplementedError("the HIP Blur takes odd kernel sizes l <= 15 with an l x l kernel (got l = %r, kernel %s)"
                                      % (l, tuple(kernel.shape)))
        k2 = kernel.reshape(l, l).double()
        c = l // 2
        # sigma from the ratio of the centre tap to its neighbour, then the kernel must BE that Gaussian
        ratio = float(k2[c, c] / k2[c, c + 1]) if l > 1 else 2.0
        self.sigma = float((1.0 / (2.0 * np.log(ratio))) ** 0.5) if ratio > 1.0 + 1e-12 else 1e6      # a box filter is the sigma -> inf Gaussian
        if not (self.sigma > 0 and (k2 - isotropic_gaussian_kernel(l, self.sigma).double()).abs().max() < 1e-7):
            raise NotImplementedError("the HIP Blur implements isotropic_gaussian_kernel(l, sigma) kernels (loss.py:153-159); this one is not")
        self.register_buffer("kernel", kernel.view(1, 1, l, l))
        if use_input_norm:
            self.register_buffer("mean", torch.Tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1))
            self.register_buffer("std", torch.Tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1))

    def forward(self, x):
        if torch.is_grad_enabled() and x.requires_grad:
            return _BlurFn.apply(x, self.use_input_norm, self.l, self.sigma)
        return E.blur_gauss(_gpu_f32(x, "Blur.forward"), self.l, self.sigma, self.use_input_norm)


class Laplacian(nn.Module):
    """loss.py:245-301: depthwise conv2d with get_laplacian_kernel2d(k), zero padding (k-1)//2."""

    def __init__(self, kernel_size):
        super().__init__()
        if not (isinstance(kernel_size, int) and 3 <= kernel_size <= 15 and kernel_size % 2 == 1):
            raise NotImplementedError("the HIP Laplacian takes an odd kernel_size in 3 .. 15 (loss.py:304 builds 3), got %r" % (kernel_size,))
        self.kernel_size = kernel_size
        self._padding = (kernel_size - 1) // 2
        self.register_buffer("kernel", get_laplacian_kernel2d(kernel_size))

    def forward(self, x):
        if torch.is_grad_enabled() and x.requires_grad:
            return _LaplacianFn.apply(x, self.kernel_size)
        return E.laplacian(_gpu_f32(x, "Laplacian.forward"), self.kernel_size)


blur_kernel = isotropic_gaussian_kernel(l=15, sigma=3.0)       # loss.py:161
blur = Blur(l=15, kernel=blur_kernel)                           # loss.py:162 (use_input_norm defaults True)
laplace_filter = Laplacian(kernel_size=3)                       # loss.py:304


class _BlurFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, use_input_norm, l=15, sigma=3.0):
        ctx.norm, ctx.l, ctx.sigma = use_input_norm, l, sigma
        return E.blur_gauss(_gpu_f32(x, "Blur.forward"), l, sigma, use_input_norm)

    @staticmethod
    def backward(ctx, dy):
        return E.blur_gauss_bwd(dy.detach().float().contiguous(), ctx.l, ctx.sigma, ctx.norm), None, None, None


class _LaplacianFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, ksize=3):
        ctx.ksize = ksize
        return E.laplacian(_gpu_f32(x, "Laplacian.forward"), ksize)

    @staticmethod
    def backward(ctx, dy):
        return E.laplacian(dy.detach().float().contiguous(), ctx.ksize), None     # self-adjoint: symmetric kernel, zero padding


def fusion_input(img, use_input_norm=True):
    """cat([img, LF(img), HF(img)], 1): what the Fusion-discriminator sees for an image
    (/root/reference/facades/network.png).  Returns NCHW fp32 (B, 9, H, W)."""
    if torch.is_grad_enabled() and img.requires_grad:      # generator's adversarial path: gradients flow through both filters
        E.require_gpu(img, "fusion_input")
        return torch.cat([img.float(), _BlurFn.apply(img, use_input_norm), _LaplacianFn.apply(img)], 1)
    x = _gpu_f32(img, "fusion_input")
    if not use_input_norm or x.shape[1] == 3:
        out = E.fusion_input_nchw(x, use_input_norm)      # the filters write into the concatenation; the Laplacian pass copies img
        if out is not None:
            return out
    return torch.cat([x, E.blur15(x, use_input_norm), E.laplacian3(x)], 1)
