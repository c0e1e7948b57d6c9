"""Host-side engine: device buffers (torch tensors as raw memory), NHWC views, thin
wrappers of the C-ABI ops and the plan recorder used by the nn.Module surface.

Nothing here computes: every function marshals pointers/sizes into
libfdgan_hip.so.  torch provides the allocator and the current HIP stream only.
"""
import ctypes as C

import torch

from . import lib as L


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_gpu(t, what):
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise RuntimeError("%s: the FD-GAN HIP path needs a tensor on an MI355X (`cuda`) device; "
                           "there is no CPU fallback" % what)


ACT_DTYPE = torch.float16       # forward activations (and the forward filter images): FD_F16
GRAD_DTYPE = torch.bfloat16     # activation gradients (and the flipped filter images): FD_BF16
_FD_DTYPE = {torch.float16: L.FD_F16, torch.bfloat16: L.FD_BF16}


class View:
    """Channel slice [c0, c0+c) of an NHWC 16-bit buffer of shape (N,H,W,C): fp16 = a forward activation, bf16 = a
    gradient (include/fdgan_hip.h, "Two 16-bit element formats")."""
    __slots__ = ("buf", "c0", "c", "fd")

    def __init__(self, buf, c0=0, c=None):
        assert buf.dtype in _FD_DTYPE and buf.dim() == 4 and buf.is_contiguous()
        n, h, w, ctot = buf.shape
        c = ctot - c0 if c is None else c
        assert 0 <= c0 and c0 + c <= ctot and c > 0
        self.buf, self.c0, self.c = buf, c0, c
        t = L.FdTensor()
        t.ptr = buf.data_ptr() + 2 * c0
        t.n, t.h, t.w, t.c = n, h, w, c
        t.stride[0], t.stride[1], t.stride[2], t.stride[3] = h * w * ctot, w * ctot, ctot, 1
        t.dtype = _FD_DTYPE[buf.dtype]
        self.fd = t

    @property
    def shape(self):
        n, h, w, _ = self.buf.shape
        return n, h, w, self.c

    def torch_nchw(self):
        """fp32 NCHW copy (debug / tests)."""
        return self.buf[..., self.c0:self.c0 + self.c].permute(0, 3, 1, 2).float().contiguous()


def nchw_f32_view(t):
    """FdTensor describing a contiguous NCHW fp32 torch tensor (strides n,h,w,c)."""
    assert t.dtype == torch.float32 and t.is_contiguous() and t.dim() == 4
    n, c, h, w = t.shape
    fd = L.FdTensor()
    fd.ptr = t.data_ptr()
    fd.n, fd.h, fd.w, fd.c = n, h, w, c
    fd.stride[0], fd.stride[1], fd.stride[2], fd.stride[3] = c * h * w, w, 1, h * w
    fd.dtype = L.FD_F32
    return fd


def new_act(n, h, w, c, device, zero=False):
    """NHWC fp16 activation buffer."""
    f = torch.zeros if zero else torch.empty
    return f((n, h, w, c), dtype=ACT_DTYPE, device=device)


def new_grad(n, h, w, c, device, zero=False):
    """NHWC bf16 gradient buffer."""
    f = torch.zeros if zero else torch.empty
    return f((n, h, w, c), dtype=GRAD_DTYPE, device=device)


def grad_like(act_buf, zero=True):
    """The bf16 gradient buffer mirroring an activation buffer (same shape, so the same views apply)."""
    f = torch.zeros if zero else torch.empty
    return f(act_buf.shape, dtype=GRAD_DTYPE, device=act_buf.device)


class PackedWeight:
    """MFMA-fragment image of one conv filter + the recipe to refresh it: fp16 for a forward convolution, bf16 for the
    flipped image a data-gradient convolution multiplies gradients with."""

    def __init__(self, param, cout, cin, k, transposed=False, flip=False, stride=1, layout=None, grad=None):
        lib = L.load()
        self.param, self.cout, self.cin, self.k = param, cout, cin, k
        self.transposed, self.flip = int(transposed), int(flip)
        # grad: the image multiplies GRADIENTS (a data-gradient convolution) -> bf16.  Default: a flipped image is one;
        # the data-gradient twin of a ConvTranspose2d 1x1 is not flipped (its IOHW weight already is the transposed
        # filter), so NetPlan.flipped_weight says grad=True explicitly.
        self.dtype = L.FD_BF16 if (flip if grad is None else grad) else L.FD_F16
        # the library names the fragment order its kernel for this conv consumes
        self.layout = lib.fdgan_conv_weight_layout(cout, cin, k, stride) if layout is None else layout
        self.nbytes = lib.fdgan_packed_weight_bytes(cout, cin, k)
        self.buf = torch.empty(self.nbytes, dtype=torch.uint8, device=param.device)

    def pack(self):
        lib = L.load()
        p = self.param.detach()
        assert p.dtype == torch.float32 and p.is_contiguous()
        L.check(lib.fdgan_pack_conv_weight(p.data_ptr(), self.cout, self.cin, self.k, self.transposed, self.flip,
                                           self.layout, self.dtype, self.buf.data_ptr(), self.nbytes, stream_ptr()),
                "pack_conv_weight")


class PackTable:
    """Every PackedWeight of a network packed by ONE launch (fdgan_pack_conv_weights): the job table is uploaded once
    and stays valid while the parameters and the packed buffers keep their addresses."""

    def __init__(self, weights):
        lib = L.load()
        self.weights = list(weights)
        jobs = (L.FdPackJob * len(self.weights))()
        first = 0
        for j, w in zip(jobs, self.weights):
            p = w.param.detach()
            assert p.dtype == torch.float32 and p.is_contiguous()
            j.w, j.packed = p.data_ptr(), w.buf.data_ptr()
            j.cout, j.cin, j.ksize, j.transposed, j.flip, j.layout = w.cout, w.cin, w.k, w.transposed, w.flip, w.layout
            j.dtype = w.dtype
            j.first_unit = first
            first += lib.fdgan_pack_units(w.cout, w.cin, w.k, w.layout)
        self.total_units = first
        raw = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8)
        self.table = raw.to(self.weights[0].param.device)
        self.ptrs = tuple(w.param.data_ptr() for w in self.weights)

    def launch(self):
        L.check(L.load().fdgan_pack_conv_weights(self.table.data_ptr(), len(self.weights), self.total_units, stream_ptr()),
                "pack_conv_weights")


def make_prologue(act=L.ACT_NONE, pool=False, mean=None, var=None, gamma=None, beta=None, eps=1e-5,
                  momentum=0.1, running_mean=None, running_var=None, nbt=None, count=0):
    p = L.FdPrologue()
    p.mean = mean.data_ptr() if mean is not None else None
    p.var = var.data_ptr() if var is not None else None
    p.gamma = gamma.data_ptr() if gamma is not None else None
    p.beta = beta.data_ptr() if beta is not None else None
    p.eps, p.act, p.pool2, p.momentum = eps, act, int(bool(pool)), momentum
    p.running_mean = running_mean.data_ptr() if running_mean is not None else None
    p.running_var = running_var.data_ptr() if running_var is not None else None
    p.num_batches_tracked = nbt.data_ptr() if nbt is not None else None
    p.count = count
    # host-side description (also keeps the tensors behind the raw pointers alive): used by the backward pass
    p._meta = dict(act=act, pool=bool(pool), mean=mean, var=var, gamma=gamma, beta=beta, eps=eps, bn=None, stats=None)
    return p


def prologue_without_side_effects(pro):
    """Same input-side fusion, but no running-statistics update: for recomputation and backward."""
    if pro is None:
        return None
    m = pro._meta
    q = make_prologue(act=m["act"], pool=m["pool"], mean=m["mean"], var=m["var"], gamma=m["gamma"], beta=m["beta"],
                      eps=m["eps"])
    q._meta.update(bn=m["bn"], stats=m["stats"], batch_stats=m.get("batch_stats", False), identity=m.get("identity", ()))
    return q


def conv_desc(k, stride=1, pad=0, e_act=L.ACT_NONE, upsample=False, cout=0, w_layout=L.WLAYOUT_CHUNK32):
    d = L.FdConvDesc()
    d.ksize, d.stride, d.pad, d.epilogue_act, d.upsample2, d.cout = k, stride, pad, e_act, int(bool(upsample)), cout
    d.w_layout = w_layout
    return d


def conv_info(x_fd, y_fd, cout, desc, pro=None):
    info = L.FdConvInfo()
    L.check(L.load().fdgan_conv2d_fwd_info(C.byref(x_fd), C.byref(y_fd), cout, C.byref(desc),
                                           C.byref(pro) if pro is not None else None, C.byref(info)), "conv2d_fwd_info")
    return info


def conv2d(x_fd, w, bias, pro, y_fd, desc, stats_buf=None, fused=None):
    """Enqueue (or record) one fused convolution.  Returns FdConvInfo when stats are produced.
    fused = (mean_ptr, var_ptr, counter_ptr, count): let the kernel's last workgroup finalize the statistics."""
    lib = L.load()
    st, info = None, None
    if stats_buf is not None:
        st = L.FdStats()
        st.partial, st.capacity_floats = stats_buf.data_ptr(), stats_buf.numel()
        if fused is not None:
            st.mean, st.var, st.counter, st.count = fused
        info = conv_info(x_fd, y_fd, desc.cout if desc.cout else y_fd.c, desc, pro)
    L.check(lib.fdgan_conv2d_fwd(C.byref(x_fd), w.buf.data_ptr() if isinstance(w, PackedWeight) else w,
                                 bias.data_ptr() if bias is not None else None,
                                 C.byref(pro) if pro is not None else None, C.byref(y_fd),
                                 C.byref(st) if st is not None else None, C.byref(desc), stream_ptr()), "conv2d_fwd")
    return info


def bn_finalize(stats_buf, info, channels, count, mean, var, c0=0):
    """mean/var are fp32 tensors; results land at [c0, c0+channels)."""
    L.check(L.load().fdgan_bn_finalize(stats_buf.data_ptr(), info.stats_rows, info.stats_cpad, channels, count,
                                       mean.data_ptr() + 4 * c0, var.data_ptr() + 4 * c0, stream_ptr()), "bn_finalize")


def to_nhwc(x_nchw_f32, view):
    n, c, h, w = x_nchw_f32.shape
    L.check(L.load().fdgan_nchw_f32_to_nhwc(x_nchw_f32.data_ptr(), n, c, h, w, C.byref(view.fd), stream_ptr()),
            "nchw_f32_to_nhwc")


def to_nchw(view, out_f32):
    L.check(L.load().fdgan_nhwc_to_nchw_f32(C.byref(view.fd), out_f32.data_ptr(), stream_ptr()),
            "nhwc_to_nchw_f32")


def maxpool2(src, dst):
    L.check(L.load().fdgan_maxpool2_nhwc(C.byref(src.fd), C.byref(dst.fd), stream_ptr()), "maxpool2_nhwc")


def maxpool2_bwd(x, dy, dx):
    L.check(L.load().fdgan_maxpool2_bwd_nhwc(C.byref(x.fd), C.byref(dy.fd), C.byref(dx.fd), stream_ptr()), "maxpool2_bwd_nhwc")


def blur15(x, use_input_norm=True):
    """x: contiguous NCHW fp32 cuda tensor -> Blur(l=15, sigma=3)(x), same shape."""
    n, c, h, w = x.shape
    y = torch.empty_like(x)
    L.check(L.load().fdgan_blur15_fwd(x.data_ptr(), y.data_ptr(), n, c, h, w, int(bool(use_input_norm)), stream_ptr()),
            "blur15_fwd")
    return y


def blur15_bwd(dy, use_input_norm=True):
    n, c, h, w = dy.shape
    tmp, dx = torch.empty_like(dy), torch.empty_like(dy)
    L.check(L.load().fdgan_blur15_bwd(dy.data_ptr(), tmp.data_ptr(), dx.data_ptr(), n, c, h, w, int(bool(use_input_norm)),
                                      stream_ptr()), "blur15_bwd")
    return dx


def blur_gauss(x, l, sigma, use_input_norm=True):
    """Blur(l, isotropic_gaussian_kernel(l, sigma))(x): odd l <= 15."""
    n, c, h, w = x.shape
    y = torch.empty_like(x)
    L.check(L.load().fdgan_blur_gauss_fwd(x.data_ptr(), y.data_ptr(), n, c, h, w, int(l), float(sigma), int(bool(use_input_norm)), stream_ptr()),
            "blur_gauss_fwd")
    return y


def blur_gauss_bwd(dy, l, sigma, use_input_norm=True):
    n, c, h, w = dy.shape
    tmp, dx = torch.empty_like(dy), torch.empty_like(dy)
    L.check(L.load().fdgan_blur_gauss_bwd(dy.data_ptr(), tmp.data_ptr(), dx.data_ptr(), n, c, h, w, int(l), float(sigma),
                                          int(bool(use_input_norm)), stream_ptr()), "blur_gauss_bwd")
    return dx


def laplacian3(x):
    n, c, h, w = x.shape
    y = torch.empty_like(x)
    L.check(L.load().fdgan_laplacian3_fwd(x.data_ptr(), y.data_ptr(), n, c, h, w, stream_ptr()), "laplacian3_fwd")
    return y


def laplacian(x, ksize):
    """Laplacian(ksize) for odd ksize in 3 .. 15; also its own adjoint (call it on dy)."""
    n, c, h, w = x.shape
    y = torch.empty_like(x)
    L.check(L.load().fdgan_laplacian_fwd(x.data_ptr(), y.data_ptr(), n, c, h, w, int(ksize), stream_ptr()), "laplacian_fwd")
    return y


def laplacian3_bwd(dy):
    n, c, h, w = dy.shape
    dx = torch.empty_like(dy)
    L.check(L.load().fdgan_laplacian3_bwd(dy.data_ptr(), dx.data_ptr(), n, c, h, w, stream_ptr()), "laplacian3_bwd")
    return dx


def fusion_input_nchw(img, use_input_norm=True):
    """cat([img, Blur(img), Laplacian(img)], 1) for a contiguous NCHW fp32 tensor, the three parts written in place by the two
    filter launches (include/fdgan_hip.h); None when the shape is outside the row-streaming kernels (nothing launched)."""
    n, c, h, w = img.shape
    out = torch.empty((n, 3 * c, h, w), dtype=torch.float32, device=img.device)
    rc = L.load().fdgan_fusion_input_nchw(img.data_ptr(), out.data_ptr(), n, c, h, w, int(bool(use_input_norm)), stream_ptr())
    if rc == L.FD_EUNSUPPORTED:
        return None
    L.check(rc, "fusion_input_nchw")
    return out


def fusion_input_nhwc(img, view, use_input_norm=True):
    """img: NCHW fp32 (n,c,h,w) -> channels [img | LF | HF] of the NHWC fp16 view (D's input)."""
    n, c, h, w = img.shape
    L.check(L.load().fdgan_fusion_input_nhwc(img.data_ptr(), n, c, h, w, C.byref(view.fd), int(bool(use_input_norm)),
                                             stream_ptr()), "fusion_input_nhwc")


def copy_nhwc(src, dst):
    L.check(L.load().fdgan_copy_nhwc(C.byref(src.fd), C.byref(dst.fd), stream_ptr()), "copy_nhwc")


def pyramid_pool4(x, weight, bias, k0, slope, y):
    """x, y: View (y: 4 channels).  weight (4, C) / bias (4,) fp32 device tensors.  include/fdgan_hip.h: fdgan_pyramid_pool4."""
    L.check(L.load().fdgan_pyramid_pool4(C.byref(x.fd), weight.data_ptr(), bias.data_ptr(), int(k0), float(slope), C.byref(y.fd),
                                         stream_ptr()), "pyramid_pool4")


def scatter_dehaze(x, tran, atp, slope, eps, window_mean, atp_out, dehaze2, cat):
    """dehaze22.py:699-715 on contiguous (N, 3, H, W) fp32 tensors; cat: View of an (N, H, W, 8) buffer."""
    n, _, h, w = x.shape
    for t in (x, tran, atp, atp_out, dehaze2):
        assert t.dtype == torch.float32 and t.is_contiguous() and tuple(t.shape) == (n, 3, h, w)
    L.check(L.load().fdgan_scatter_dehaze(x.data_ptr(), tran.data_ptr(), atp.data_ptr(), n, h, w, float(slope), float(eps),
                                          window_mean.data_ptr(), atp_out.data_ptr(), dehaze2.data_ptr(), C.byref(cat.fd), stream_ptr()),
            "scatter_dehaze")


def maxpool3s2(x, pro, y, stats_buf=None):
    """y = MaxPool2d(3, 2, 1)(act(bn(x))); returns the number of statistics rows written to stats_buf (0 without one)."""
    rows = C.c_int64(0)
    L.check(L.load().fdgan_maxpool3s2_nhwc(C.byref(x.fd), C.byref(pro) if pro is not None else None, C.byref(y.fd),
                                           stats_buf.data_ptr() if stats_buf is not None else None,
                                           stats_buf.numel() if stats_buf is not None else 0, C.byref(rows), stream_ptr()), "maxpool3s2_nhwc")
    return rows.value


def bn_dropout(x, mean, var, gamma, beta, eps, mask, y):
    """y = mask[n][c] * bn(x) on NHWC fp16 views; any of (mean, var) / gamma / beta / mask may be None."""
    ptr = lambda t: t.data_ptr() if t is not None else None
    L.check(L.load().fdgan_bn_dropout_nhwc(C.byref(x.fd), ptr(mean), ptr(var), ptr(gamma), ptr(beta), float(eps), ptr(mask),
                                           C.byref(y.fd), stream_ptr()), "bn_dropout_nhwc")


def maxpool3s2_bwd(x, pro, dy, da):
    """da (written) = gradient w.r.t. act(bn(x)) of MaxPool2d(3, 2, 1): dy routed to each window's first maximum."""
    L.check(L.load().fdgan_maxpool3s2_bwd(C.byref(x.fd), C.byref(pro) if pro is not None else None, C.byref(dy.fd), C.byref(da.fd), stream_ptr()),
            "maxpool3s2_bwd")


def pyramid_pool4_bwd(x, weight, bias, k0, slope, dy, dx):
    """dx += the head's input gradient; returns (dweight (4, C), dbias (4,)) fp32."""
    n, h, w, c = x.shape
    tiles = n * (h // k0) * (w // k0)
    dwp = torch.empty((tiles, 4, c), dtype=torch.float32, device=x.buf.device)
    dbp = torch.empty((tiles, 4), dtype=torch.float32, device=x.buf.device)
    t = C.c_int64(0)
    L.check(L.load().fdgan_pyramid_pool4_bwd(C.byref(x.fd), weight.data_ptr(), bias.data_ptr(), int(k0), float(slope), C.byref(dy.fd), C.byref(dx.fd),
                                             dwp.data_ptr(), dbp.data_ptr(), C.byref(t), stream_ptr()), "pyramid_pool4_bwd")
    assert t.value == tiles
    return dwp.sum(0), dbp.sum(0)


def bn_dropout_bwd(x, mean, var, gamma, eps, mask, dy, dx):
    """Backward of y = mask * bn(x) (batch statistics): dx written; returns (dgamma, dbeta) or (None, None) without a norm."""
    ptr = lambda t: t.data_ptr() if t is not None else None
    dg = db = None
    if mean is not None:
        dg = torch.empty(x.c, dtype=torch.float32, device=x.buf.device)
        db = torch.empty(x.c, dtype=torch.float32, device=x.buf.device)
    L.check(L.load().fdgan_bn_dropout_bwd(C.byref(x.fd), ptr(mean), ptr(var), ptr(gamma), float(eps), ptr(mask), C.byref(dy.fd), C.byref(dx.fd),
                                          ptr(dg), ptr(db), stream_ptr()), "bn_dropout_bwd")
    return dg, db


def scatter_dehaze_bwd(x, tran, atp, window_mean, slope, eps, g_dehaze2, g_atp, g_cat):
    """Gradients (d_tran, d_atp), both (N, 3, H, W) fp32, of dehaze22.py:699-715; g_dehaze2 / g_atp: fp32 NCHW or None, g_cat:
    gradient View of the refine input (its channels 0-2 are J's) or None."""
    n, _, h, w = x.shape
    d_tran, d_atp = torch.empty_like(tran), torch.empty_like(atp)
    scratch = torch.empty(n * 3 * max(w // h, 1) * h, dtype=torch.float32, device=x.device)
    ptr = lambda t: t.data_ptr() if t is not None else None
    for t in (g_dehaze2, g_atp):
        assert t is None or (t.dtype == torch.float32 and t.is_contiguous() and tuple(t.shape) == (n, 3, h, w))
    L.check(L.load().fdgan_scatter_dehaze_bwd(x.data_ptr(), tran.data_ptr(), atp.data_ptr(), window_mean.data_ptr(), n, h, w, float(slope),
                                              float(eps), ptr(g_dehaze2), ptr(g_atp), C.byref(g_cat.fd) if g_cat is not None else None,
                                              d_tran.data_ptr(), d_atp.data_ptr(), scratch.data_ptr(), scratch.numel(), stream_ptr()),
            "scatter_dehaze_bwd")
    return d_tran, d_atp


class StridedView:
    """An NHWC 16-bit view that steps over rows and pixels of its buffer: channels [c0, c0 + c) of the pixels
    (y0 + sy i, x0 + sx j), i < h, j < w -- the output positions of one parity of a stride-2 transposed convolution."""
    __slots__ = ("buf", "c0", "c", "fd", "geom")

    def __init__(self, buf, c0, c, y0, x0, sy, sx, h, w):
        assert buf.dtype in _FD_DTYPE and buf.dim() == 4 and buf.is_contiguous()
        n, hh, ww, ctot = buf.shape
        assert y0 + sy * (h - 1) < hh and x0 + sx * (w - 1) < ww and c0 + c <= ctot
        self.buf, self.c0, self.c = buf, c0, c
        self.geom = (y0, x0, sy, sx, h, w)
        t = L.FdTensor()
        t.ptr = buf.data_ptr() + 2 * ((y0 * ww + x0) * ctot + c0)
        t.n, t.h, t.w, t.c = n, h, w, c
        t.stride[0], t.stride[1], t.stride[2], t.stride[3] = hh * ww * ctot, sy * ww * ctot, sx * ctot, 1
        t.dtype = _FD_DTYPE[buf.dtype]
        self.fd = t

    @property
    def shape(self):
        return self.fd.n, self.fd.h, self.fd.w, self.c

    def torch_nchw(self):
        """fp32 NCHW copy (debug / tests)."""
        y0, x0, sy, sx, h, w = self.geom
        return self.buf[:, y0:y0 + sy * h:sy, x0:x0 + sx * w:sx, self.c0:self.c0 + self.c].permute(0, 3, 1, 2).float().contiguous()


class Plan:
    """RAII wrapper of FdPlan: `with plan.record(): <op calls>` then plan.launch()."""

    def __init__(self):
        self.lib = L.load()
        self.h = C.c_void_p(self.lib.fdgan_plan_create())
        if not self.h:
            raise MemoryError("fdgan_plan_create")
        self.graph = False

    def close(self):
        if self.h:
            self.lib.fdgan_plan_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    class _Rec:
        def __init__(self, plan):
            self.plan = plan

        def __enter__(self):
            L.check(self.plan.lib.fdgan_plan_begin(self.plan.h), "plan_begin")

        def __exit__(self, *exc):
            L.check(self.plan.lib.fdgan_plan_end(self.plan.h), "plan_end")
            return False

    def record(self):
        return Plan._Rec(self)

    def launch(self):
        L.check(self.lib.fdgan_plan_launch(self.h, stream_ptr()), "plan_launch")

    def instantiate_graph(self):
        L.check(self.lib.fdgan_plan_instantiate_graph(self.h, stream_ptr()), "plan_instantiate_graph")
        self.graph = True

    def __len__(self):
        return int(self.lib.fdgan_plan_num_launches(self.h))

    def kernel_names(self):
        return [self.lib.fdgan_plan_kernel_name(self.h, i).decode() for i in range(len(self))]

    def time_launches(self, idx):
        arr = (C.c_int64 * len(idx))(*sorted(idx))
        L.check(self.lib.fdgan_plan_time_launches(self.h, arr, len(idx)), "plan_time_launches")

    def read_timing(self):
        tot, cnt = C.c_double(0.0), C.c_int64(0)
        L.check(self.lib.fdgan_plan_read_timing(self.h, C.byref(tot), C.byref(cnt)), "plan_read_timing")
        return tot.value, cnt.value

    def profile(self):
        n = len(self)
        ms = (C.c_float * n)()
        L.check(self.lib.fdgan_plan_profile(self.h, stream_ptr(), ms, n), "plan_profile")
        return list(ms)


# ---- backward ------------------------------------------------------------------------------------
def conv_bwd_weight(x_fd, pro, dy_fd, desc, dw, dbias=None, ws=None, accumulate=False):
    """dw: fp32 tensor shaped like the Conv2d weight (cout, cin, k, k); dbias: fp32 (cout,) or None;
    ws: fp32 workspace for the split-K partials (None: one pass over all pixels per workgroup)."""
    L.check(L.load().fdgan_conv2d_bwd_weight(C.byref(x_fd), C.byref(pro) if pro is not None else None, C.byref(dy_fd),
                                             C.byref(desc), dw.data_ptr(), dbias.data_ptr() if dbias is not None else None,
                                             ws.data_ptr() if ws is not None else None, ws.numel() if ws is not None else 0,
                                             int(bool(accumulate)), stream_ptr()), "conv2d_bwd_weight")


def conv_bwd_weight_job(x_fd, pro, dy_fd, desc, dw, ws, defer, accumulate=False):
    """conv_bwd_weight that also returns the description (lib.FdTrReduceJob) of the kernel's final reduction over the partial sums
    in `ws`, or None when the kernel chosen for the shape has none; defer: do not launch that reduction (TrReduceTable does)."""
    job = L.FdTrReduceJob()
    L.check(L.load().fdgan_conv2d_bwd_weight_job(C.byref(x_fd), C.byref(pro) if pro is not None else None, C.byref(dy_fd),
                                                 C.byref(desc), dw.data_ptr(), None, ws.data_ptr(), ws.numel(), int(bool(accumulate)),
                                                 C.byref(job), int(bool(defer)), stream_ptr()), "conv2d_bwd_weight_job")
    return job if job.part else None


class TrReduceTable:
    """The deferred reductions of a backward walk's row-walking weight gradients as ONE launch (fdgan_wgrad_tr_reduce_batch).
    jobs: the lib.FdTrReduceJob structures conv_bwd_weight_job(defer=True) returned; keep: the tensors they point into."""

    def __init__(self, jobs, keep, device):
        self.keep, self.n = list(keep), len(jobs)
        tab = (L.FdTrReduceJob * len(jobs))()
        first = 0
        for t, j in zip(tab, jobs):
            C.memmove(C.byref(t), C.byref(j), C.sizeof(L.FdTrReduceJob))
            t.first_group = first
            first += j.groups
        self.groups = first
        self.table = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).to(device)
        self.key = tuple((j.part, j.out, j.item_stride, j.items, j.accumulate) for j in jobs)

    def launch(self):
        L.check(L.load().fdgan_wgrad_tr_reduce_batch(self.table.data_ptr(), self.n, self.groups, stream_ptr()), "wgrad_tr_reduce_batch")


def bn_act_bwd(da_fd, x_fd, pro, ws=None, dx_fd=None, dx_store=False):
    """In place: da <- da * act'(bn(x)).  With a norm in `pro`, also fills `ws` with the partial sums and
    returns (rows, cpad) for bn_bwd_finalize.  dx_fd (pooled prologues only): dx += gamma * rstd * dpre in the same pass
    (dx_store: dx = ..., the buffer's first writer of a walk)."""
    rows, cpad = C.c_int64(0), C.c_int64(0)
    L.check(L.load().fdgan_bn_act_bwd_dx(C.byref(da_fd), C.byref(x_fd), C.byref(pro) if pro is not None else None,
                                         C.byref(dx_fd) if dx_fd is not None else None, int(bool(dx_store)),
                                         ws.data_ptr() if ws is not None else None, ws.numel() if ws is not None else 0,
                                         C.byref(rows), C.byref(cpad), stream_ptr()), "bn_act_bwd")
    return rows.value, cpad.value


def bn_bwd_finalize(ws, rows, cpad, channels, dgamma, dbeta, accumulate=False, sink_dgamma=None, sink_dbeta=None):
    """(dgamma, dbeta) <- the reduced partials of bn_act_bwd; sink_*: fp32 gradient buffers the sums are also added to."""
    L.check(L.load().fdgan_bn_bwd_finalize_sink(ws.data_ptr(), rows, cpad, channels, dgamma.data_ptr(), dbeta.data_ptr(),
                                                int(bool(accumulate)),
                                                sink_dgamma.data_ptr() if sink_dgamma is not None else None,
                                                sink_dbeta.data_ptr() if sink_dbeta is not None else None, stream_ptr()),
            "bn_bwd_finalize")


def bn_bwd_finalize_raw(ws, rows, cpad, channels, mean, var, eps, dgamma, dbeta, sink_dgamma=None, sink_dbeta=None, scratch=None):
    """(dgamma, dbeta) from the (sum dpre, sum dpre * x) partials of conv_bwd_data.  scratch: fp32 tensor of at least
    64 * cpad elements enabling the two-level reduction of many rows."""
    L.check(L.load().fdgan_bn_bwd_finalize_raw(ws.data_ptr(), rows, cpad, channels, mean.data_ptr(), var.data_ptr(), eps,
                                               dgamma.data_ptr(), dbeta.data_ptr(),
                                               sink_dgamma.data_ptr() if sink_dgamma is not None else None,
                                               sink_dbeta.data_ptr() if sink_dbeta is not None else None,
                                               scratch.data_ptr() if scratch is not None else None,
                                               scratch.numel() if scratch is not None else 0, stream_ptr()),
            "bn_bwd_finalize_raw")


def bn_bwd_finalize_coef(ws, rows, cpad, channels, pro, count, bsum, csum, sink_dgamma=None, sink_dbeta=None, scratch=None, store=False):
    """bn_bwd_finalize_raw + bn_bwd_coef in one launch (no dgamma / dbeta temporaries).  store: bsum / csum are written, not added to."""
    L.check(L.load().fdgan_bn_bwd_finalize_coef(ws.data_ptr(), rows, cpad, channels, C.byref(pro), count,
                                                sink_dgamma.data_ptr() if sink_dgamma is not None else None,
                                                sink_dbeta.data_ptr() if sink_dbeta is not None else None,
                                                bsum.data_ptr(), csum.data_ptr(),
                                                scratch.data_ptr() if scratch is not None else None,
                                                scratch.numel() if scratch is not None else 0, int(bool(store)), stream_ptr()),
            "bn_bwd_finalize_coef")


def conv_bwd_data(dy_fd, pw_flipped, fwd_x_fd, fwd_pro, dpre_fd, desc, ws=None, accumulate=False):
    """dpre <- conv^T(dy, W) * act'(bn(fwd_x)) (stride-1 convs); with a norm in fwd_pro fills `ws` and returns
    (rows, cpad) for bn_bwd_finalize_raw.  accumulate: dpre_fd is the gradient buffer of fwd_x; 1 / True: += gamma * rstd *
    dpre, 2: = gamma * rstd * dpre (sole consumer)."""
    rows, cpad = C.c_int64(0), C.c_int64(0)
    L.check(L.load().fdgan_conv2d_bwd_data(C.byref(dy_fd), pw_flipped.buf.data_ptr(), C.byref(fwd_x_fd),
                                           C.byref(fwd_pro) if fwd_pro is not None else None, C.byref(dpre_fd),
                                           int(accumulate), ws.data_ptr() if ws is not None else None, ws.numel() if ws is not None else 0,
                                           C.byref(rows), C.byref(cpad), C.byref(desc), stream_ptr()), "conv2d_bwd_data")
    return rows.value, cpad.value


def conv1x1_bwd_data_weight(dy_fd, pw_flipped, fwd_x_fd, fwd_pro, dpre_fd, ws_bn, accumulate, wgrad_ws, dw, dw_accumulate=True,
                            dy_affine=None):
    """The bottleneck's data gradient (as conv_bwd_data) and weight gradient in one pass; returns (rows, cpad), or None when
    the shape is outside the fused kernel (nothing launched).  dy_affine = (x_fd, B, C): dy + B * x + C is used instead of dy
    (the pending linear part of the BatchNorm backward of dy's producer).  dw=None: the per-slot partials stay in wgrad_ws for a
    later reduce_batch; the return value is then (rows, cpad, nsplit)."""
    rows, cpad, nsplit = C.c_int64(0), C.c_int64(0), C.c_int64(0)
    assert dw is None or (dw.dtype == torch.float32 and dw.is_contiguous())
    rc = L.load().fdgan_conv1x1_bwd_data_weight(C.byref(dy_fd), pw_flipped.buf.data_ptr(), C.byref(fwd_x_fd),
                                                C.byref(fwd_pro) if fwd_pro is not None else None, C.byref(dpre_fd), int(accumulate),
                                                ws_bn.data_ptr() if ws_bn is not None else None, ws_bn.numel() if ws_bn is not None else 0,
                                                C.byref(rows), C.byref(cpad), wgrad_ws.data_ptr(), wgrad_ws.numel(),
                                                dw.data_ptr() if dw is not None else None,
                                                int(bool(dw_accumulate)), C.byref(dy_affine[0]) if dy_affine is not None else None,
                                                dy_affine[1].data_ptr() if dy_affine is not None else None,
                                                dy_affine[2].data_ptr() if dy_affine is not None else None, C.byref(nsplit), stream_ptr())
    if rc == L.FD_EUNSUPPORTED:
        return None
    L.check(rc, "conv1x1_bwd_data_weight")
    return (rows.value, cpad.value) if dw is not None else (rows.value, cpad.value, nsplit.value)


class ReduceTable:
    """Every deferred [nsplit][numel] -> out reduction of a backward walk as ONE launch (fdgan_wgrad_reduce_batch).  jobs:
    list of (partials tensor, out tensor, numel, nsplit, accumulate); the table is uploaded once and stays valid while those
    tensors keep their addresses."""

    def __init__(self, jobs, device):
        self.keep = list(jobs)
        tab = (L.FdReduceJob * len(jobs))()
        first = 0
        for t, (part, out, numel, nsplit, acc) in zip(tab, jobs):
            assert part.dtype == torch.float32 and out.dtype == torch.float32 and part.numel() >= numel * nsplit and out.numel() >= numel
            t.part, t.out, t.numel, t.nsplit, t.accumulate, t.first_group = part.data_ptr(), out.data_ptr(), numel, nsplit, int(bool(acc)), first
            first += (numel + 63) // 64
        self.groups = first
        self.table = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).to(device)
        self.key = tuple((p.data_ptr(), o.data_ptr(), n, s, a) for p, o, n, s, a in jobs)

    def launch(self):
        L.check(L.load().fdgan_wgrad_reduce_batch(self.table.data_ptr(), len(self.keep), self.groups, stream_ptr()), "wgrad_reduce_batch")


def bn_bwd_coef(dgamma, dbeta, pro, channels, count, bsum, csum):
    """bsum / csum (fp32 views) += this layer's (B, C) of dx = A*dpre + B*x + C."""
    L.check(L.load().fdgan_bn_bwd_coef(dgamma.data_ptr(), dbeta.data_ptr(), C.byref(pro), channels, count, bsum.data_ptr(),
                                       csum.data_ptr(), stream_ptr()), "bn_bwd_coef")


def affine_accumulate(x_fd, bsum, csum, dx_fd, out_fd=None):
    """dx += bsum * x + csum per channel; with out_fd: out = dx + bsum * x + csum, dx untouched."""
    if out_fd is None:
        L.check(L.load().fdgan_affine_accumulate(C.byref(x_fd), bsum.data_ptr(), csum.data_ptr(), C.byref(dx_fd), stream_ptr()),
                "affine_accumulate")
    else:
        L.check(L.load().fdgan_affine_accumulate_out(C.byref(x_fd), bsum.data_ptr(), csum.data_ptr(), C.byref(dx_fd), C.byref(out_fd),
                                                     stream_ptr()), "affine_accumulate_out")


def bn_bwd_apply(dpre_fd, x_fd, pro, dgamma, dbeta, dx_fd, accumulate=False):
    L.check(L.load().fdgan_bn_bwd_apply(C.byref(dpre_fd), C.byref(x_fd), C.byref(pro), dgamma.data_ptr(), dbeta.data_ptr(),
                                        C.byref(dx_fd), int(bool(accumulate)), stream_ptr()), "bn_bwd_apply")


def conv_bwd_data_direct(dy_fd, weight, desc, dx):
    """dx: contiguous NCHW fp32 (n, cin, h, w), overwritten; weight: fp32 (cout, cin, k, k)."""
    n, cin, h, w = dx.shape
    L.check(L.load().fdgan_conv2d_bwd_data_direct(C.byref(dy_fd), weight.data_ptr(), weight.shape[0], cin, C.byref(desc),
                                                  dx.data_ptr(), n, h, w, stream_ptr()), "conv2d_bwd_data_direct")


def conv_bwd_data_direct_nhwc(dy_fd, weight, desc, dx_view, cin):
    """dx_view: NHWC bf16 view written for channels [0, cin); weight: fp32 (cout, cin, k, k)."""
    L.check(L.load().fdgan_conv2d_bwd_data_direct_nhwc(C.byref(dy_fd), weight.data_ptr(), weight.shape[0], cin, C.byref(desc),
                                                       C.byref(dx_view.fd), stream_ptr()), "conv2d_bwd_data_direct_nhwc")


def out_act_bwd(dout, out, act, g_view):
    """g_view (NHWC bf16) <- dout * f'(out) for contiguous NCHW fp32 dout / out."""
    n, c, h, w = out.shape
    L.check(L.load().fdgan_out_act_bwd(dout.data_ptr(), out.data_ptr(), n, c, h, w, act, C.byref(g_view.fd), stream_ptr()),
            "out_act_bwd")


def fill_zero(t):
    """t.zero_() as a launch of the library (recordable into a plan).  t: a contiguous tensor, or a 2-D row-slice view
    (`coef[:, lo:hi]`) of 4-byte elements."""
    if t.dim() == 2 and not t.is_contiguous():
        assert t.stride(1) == 1 and t.element_size() == 4
        rows, rb, rs = t.shape[0], t.shape[1] * 4, t.stride(0) * 4
    else:
        assert t.is_contiguous() and (t.numel() * t.element_size()) % 4 == 0
        rows, rb, rs = 1, t.numel() * t.element_size(), 0
    if rb:
        L.check(L.load().fdgan_fill_zero(t.data_ptr(), rb, rows, rs, stream_ptr()), "fill_zero")


class ZeroTable:
    """Several whole buffers zeroed by ONE launch (fdgan_fill_zero_many); the device table stays valid while they keep their addresses."""

    def __init__(self, tensors):
        self.keep = list(tensors)
        tab = (L.FdZeroJob * len(self.keep))()
        first = 0
        for j, t in zip(tab, self.keep):
            nb = t.numel() * t.element_size()
            assert t.is_contiguous() and nb % 16 == 0 and t.data_ptr() % 16 == 0
            j.ptr, j.bytes, j.first_group = t.data_ptr(), nb, first
            first += (nb + 16383) // 16384
        self.groups = first
        self.table = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).to(self.keep[0].device)

    def launch(self):
        L.check(L.load().fdgan_fill_zero_many(self.table.data_ptr(), len(self.keep), self.groups, stream_ptr()), "fill_zero_many")


def add_transposed(dst, src):
    """dst (cols, rows) += src (rows, cols) transposed; fp32, contiguous."""
    assert dst.dtype == torch.float32 and src.dtype == torch.float32 and dst.is_contiguous() and src.is_contiguous()
    rows, cols = src.shape[0], src.numel() // src.shape[0]
    assert dst.numel() == src.numel()
    L.check(L.load().fdgan_add_transposed_f32(dst.data_ptr(), src.data_ptr(), rows, cols, stream_ptr()), "add_transposed_f32")


def mul_mask(mask, dst, up2=False):
    """dst (activation or gradient View) *= mask (fp16 View holding 0 or 1 / (1 - p)); up2: dst is the x2 upsampled image of the mask's extent."""
    L.check(L.load().fdgan_mul_mask_nhwc(C.byref(mask.fd), C.byref(dst.fd), int(bool(up2)), stream_ptr()), "mul_mask_nhwc")


GRAD_ADD, GRAD_UNPOOL, GRAD_SUMPOOL, GRAD_RELU_MASK, GRAD_LEAKY_MASK = 0, 1, 2, 3, 4


def grad_ew(mode, src, dst, ref=None):
    L.check(L.load().fdgan_grad_ew(mode, C.byref(src.fd), C.byref(ref.fd) if ref is not None else None, C.byref(dst.fd),
                                   stream_ptr()), "grad_ew")


class cu_budget:
    """`with cu_budget(n):` -- the persistent kernels launched (or recorded) inside size their grids for n CUs (a stream created with
    a CU mask: tools/cu_mask_sweep.py); 0 / None: the whole device."""

    def __init__(self, ncu):
        self.ncu = int(ncu or 0)

    def __enter__(self):
        self.prev = L.load().fdgan_set_cu_budget(self.ncu)
        return self

    def __exit__(self, *exc):
        L.load().fdgan_set_cu_budget(self.prev)
        return False


def kernel_timer_arm(name=None, stride=1, max_samples=4096):
    """Bracket every `stride`-th launch named `name` (None: every launch) with hipEvents on its stream."""
    L.check(L.load().fdgan_kernel_timer_arm(name.encode() if name else None, stride, max_samples), "kernel_timer_arm")


def kernel_timer_read(capacity=65536):
    """-> (list of (launch index among the matching launches, ms, launcher name), number of matching launches)."""
    n, seen = C.c_int(0), C.c_int(0)
    idx, ms, names = (C.c_int * capacity)(), (C.c_float * capacity)(), C.create_string_buffer(48 * capacity)
    L.check(L.load().fdgan_kernel_timer_read(capacity, C.byref(n), C.byref(seen), idx, ms, names), "kernel_timer_read")
    out = [(idx[i], ms[i], names.raw[48 * i:48 * i + 48].split(b"\0", 1)[0].decode()) for i in range(n.value)]
    return out, seen.value
