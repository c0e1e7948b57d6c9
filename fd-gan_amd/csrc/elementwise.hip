// elementwise.hip -- the small HBM-bound kernels around the convolutions:
// weight packing, train-mode BatchNorm statistics finalisation, layout conversion,
// channel-slice copy.  gfx950; 16-byte vector accesses wherever the layout allows.
#include "common.h"

// ---------------------------------------------------------------------------------
// weight packing: fp32 OIHW / IOHW -> fp16 (forward image) or bf16 (flipped image of the data gradient) MFMA fragment order
//   packed[((chunk*KK + tap)*ntile + tile)*512 + lane*8 + e]
//     = W[cout = tile*16 + (lane&15)][cin = chunk*32 + (lane>>4)*8 + e][tap]
// ---------------------------------------------------------------------------------
struct PackArgs {
  const float* w;
  unsigned short* out;
  int cout, cin, kk, ks, ntile, nchunk, transposed, flip, layout, dtype;
  long long nunits;
};

__global__ void pack_weight_kernel(PackArgs a) {
  const long long u = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= a.nunits) return;
  const int lane = (int)(u & 63);
  long long r = u >> 6;
  const int tile = (int)(r % a.ntile);
  r /= a.ntile;
  const int co = tile * 16 + (lane & 15);
  int tap, ci0;
  if (a.layout == FD_WLAYOUT_X64) {  // r = ks*2 + j; lane group g owns channels ks*64 + g*16 + j*8 ..
    tap = 0;
    ci0 = (int)(r >> 1) * 64 + (lane >> 4) * 16 + (int)(r & 1) * 8;
  } else {
    tap = (int)(r % a.kk);
    ci0 = (int)(r / a.kk) * 32 + (lane >> 4) * 8;
  }
  f32x8 v;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int ci = ci0 + e;
    // logical filter F[co][ci][tap]; source tensor S is OIHW (or IOHW if transposed).
    // flip: F[co][ci][tap] = S'[ci][co][kk-1-tap] (data-gradient filter)
    // (the eight loads are unconditional -- element 0 for a padding position, zeroed after: inside `if (in range)` each of them was
    // waited for before the next was issued)
    const bool ok = co < a.cout && ci < a.cin;
    int o = co, i = ci, t = tap;
    if (a.flip) {
      o = ci;
      i = co;
      t = a.kk - 1 - tap;
    }
    const int d0 = a.flip ? a.cin : a.cout, d1 = a.flip ? a.cout : a.cin;  // S' logical dims (O', I')
    const long long idx = a.transposed ? ((long long)i * d0 + o) * a.kk + t : ((long long)o * d1 + i) * a.kk + t;
    const float x = a.w[ok ? idx : 0];
    v[e] = ok ? x : 0.f;
  }
  *reinterpret_cast<u32x4*>(a.out + u * 8) = a.dtype == FD_F16 ? fd_pk8<FmtA>(v) : fd_pk8<FmtG>(v);
}

extern "C" size_t fdgan_packed_weight_bytes(int cout, int cin, int ksize) {
  if (cout <= 0 || cin <= 0 || ksize <= 0) return 0;
  const size_t ntile = (cout + 15) / 16, nchunk = (cin + 31) / 32, nks = (cin + 63) / 64;
  if (ksize == 1) return nks * 2 * ntile * 1024;  // X64 (>= CHUNK32 for a 1x1)
  return nchunk * (size_t)ksize * ksize * ntile * 1024;
}

extern "C" int fdgan_pack_conv_weight(const float* w, int cout, int cin, int ksize, int transposed, int flip,
                                      int layout, int dtype, void* packed, size_t packed_bytes, FdStream stream) {
  FD_REQUIRE(w && packed, "pack_conv_weight: NULL pointer");
  FD_REQUIRE(dtype == FD_F16 || dtype == FD_BF16, "pack_conv_weight: dtype must be FD_F16 (forward image) or FD_BF16 (gradient-side image)");
  FD_REQUIRE(cout > 0 && cin > 0 && ksize > 0, "pack_conv_weight: bad shape");
  FD_REQUIRE(layout == FD_WLAYOUT_CHUNK32 || (layout == FD_WLAYOUT_X64 && ksize == 1),
             "pack_conv_weight: layout %d not valid for ksize %d", layout, ksize);
  FD_REQUIRE(packed_bytes >= fdgan_packed_weight_bytes(cout, cin, ksize), "pack_conv_weight: buffer too small");
  FD_REQUIRE(((uintptr_t)packed & 15) == 0, "pack_conv_weight: packed must be 16-byte aligned");
  PackArgs a;
  a.w = w;
  a.out = static_cast<unsigned short*>(packed);
  a.cout = cout;
  a.cin = cin;
  a.ks = ksize;
  a.kk = ksize * ksize;
  a.ntile = (cout + 15) / 16;
  a.nchunk = (cin + 31) / 32;
  a.transposed = transposed;
  a.flip = flip;
  a.layout = layout;
  a.dtype = dtype;
  a.nunits = layout == FD_WLAYOUT_X64 ? (long long)((cin + 63) / 64) * 2 * a.ntile * 64
                                      : (long long)a.nchunk * a.kk * a.ntile * 64;
  const unsigned nb = (unsigned)((a.nunits + 255) / 256);
  return fd_launch(&pack_weight_kernel, "pack_weight", dim3(nb), dim3(256), 0, a, static_cast<hipStream_t>(stream));
}

// Batched form: every filter of a network (both orientations) in ONE launch.  The job table lives in device memory; a unit
// (16 bytes of packed output) finds its job by bisection over the jobs' first units.
struct PackJobsArgs {
  const FdPackJob* jobs;
  int njobs;
  long long nunits;
};

__global__ void pack_weights_kernel(PackJobsArgs b) {
  const long long gu = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gu >= b.nunits) return;
  int lo = 0, hi = b.njobs - 1;
  while (lo < hi) {   // last job with first_unit <= gu
    const int mid = (lo + hi + 1) >> 1;
    if (b.jobs[mid].first_unit <= gu) lo = mid;
    else hi = mid - 1;
  }
  const FdPackJob j = b.jobs[lo];
  const long long u = gu - j.first_unit;
  const int kk = j.ksize * j.ksize, ntile = (j.cout + 15) / 16;
  const int lane = (int)(u & 63);
  long long r = u >> 6;
  const int tile = (int)(r % ntile);
  r /= ntile;
  const int co = tile * 16 + (lane & 15);
  int tap, ci0;
  if (j.layout == FD_WLAYOUT_X64) {
    tap = 0;
    ci0 = (int)(r >> 1) * 64 + (lane >> 4) * 16 + (int)(r & 1) * 8;
  } else {
    tap = (int)(r % kk);
    ci0 = (int)(r / kk) * 32 + (lane >> 4) * 8;
  }
  f32x8 v;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int ci = ci0 + e;
    const bool ok = co < j.cout && ci < j.cin;      // (unconditional loads: see pack_weight_kernel)
    int o = co, i = ci, t = tap;
    if (j.flip) {
      o = ci;
      i = co;
      t = kk - 1 - tap;
    }
    const int d0 = j.flip ? j.cin : j.cout, d1 = j.flip ? j.cout : j.cin;
    const long long idx = j.transposed ? ((long long)i * d0 + o) * kk + t : ((long long)o * d1 + i) * kk + t;
    const float x = j.w[ok ? idx : 0];
    v[e] = ok ? x : 0.f;
  }
  *reinterpret_cast<u32x4*>(static_cast<unsigned short*>(j.packed) + u * 8) = j.dtype == FD_F16 ? fd_pk8<FmtA>(v) : fd_pk8<FmtG>(v);
}

extern "C" int64_t fdgan_pack_units(int cout, int cin, int ksize, int layout) {
  if (cout <= 0 || cin <= 0 || ksize <= 0) return 0;
  const long long ntile = (cout + 15) / 16;
  return layout == FD_WLAYOUT_X64 ? (long long)((cin + 63) / 64) * 2 * ntile * 64
                                  : (long long)((cin + 31) / 32) * ksize * ksize * ntile * 64;
}

extern "C" int fdgan_pack_conv_weights(const FdPackJob* jobs_device, int64_t njobs, int64_t total_units, FdStream stream) {
  FD_REQUIRE(jobs_device && njobs > 0 && njobs < (1 << 20) && total_units > 0, "pack_conv_weights: empty job table");
  FD_REQUIRE(total_units < (1ll << 31) * 256, "pack_conv_weights: too many units");
  PackJobsArgs b{jobs_device, (int)njobs, (long long)total_units};
  const unsigned nb = (unsigned)((total_units + 255) / 256);
  return fd_launch(&pack_weights_kernel, "pack_weights", dim3(nb), dim3(256), 0, b, static_cast<hipStream_t>(stream));
}

// ---------------------------------------------------------------------------------
// BatchNorm statistics: partial[rows][cpad][2] -> mean, biased var
// (nn.BatchNorm2d train mode).  One workgroup per 32 channels;
// 8 row groups x 32 channels; accumulation in fp64.
// ---------------------------------------------------------------------------------
struct BnFinArgs {
  const float* partial;
  long long rows, cpad, channels;
  double inv_count;
  float *mean, *var;
};

// One workgroup per 32 channels, 32 row groups x 32 channels: with <= 512 partial rows every thread has
// all of its (<= 16) loads in flight at once, so the kernel is two memory latencies long.
__global__ __launch_bounds__(1024) void bn_finalize_kernel(BnFinArgs a) {
  __shared__ double sh[2][32][33];
  const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const long long c = (long long)blockIdx.x * 32 + cl;
  double s1 = 0.0, s2 = 0.0;
  if (c < a.channels) {
    const float2* p = reinterpret_cast<const float2*>(a.partial) + c;
    long long r = rg;
    for (; r + 224 < a.rows; r += 256) {  // 8 independent loads in flight
      float2 v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = p[(r + 32 * i) * a.cpad];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        s1 += (double)v[i].x;
        s2 += (double)v[i].y;
      }
    }
    for (; r < a.rows; r += 32) {
      const float2 v = p[r * a.cpad];
      s1 += v.x;
      s2 += v.y;
    }
  }
  sh[0][rg][cl] = s1;
  sh[1][rg][cl] = s2;
  __syncthreads();
  if (rg == 0 && c < a.channels) {
    double t1 = 0.0, t2 = 0.0;
#pragma unroll
    for (int g = 0; g < 32; ++g) {
      t1 += sh[0][g][cl];
      t2 += sh[1][g][cl];
    }
    const double mean = t1 * a.inv_count;
    double var = t2 * a.inv_count - mean * mean;
    if (var < 0.0) var = 0.0;
    a.mean[c] = (float)mean;
    a.var[c] = (float)var;
  }
}

extern "C" int fdgan_bn_finalize(const float* partial, int64_t rows, int64_t cpad, int64_t channels,
                                 int64_t count, float* mean, float* var, FdStream stream) {
  FD_REQUIRE(partial && mean && var, "bn_finalize: NULL pointer");
  FD_REQUIRE(rows > 0 && channels > 0 && cpad >= channels && count > 0, "bn_finalize: bad sizes");
  {  // measurement aid (tuning builds; results wrong): see launch_sum_finalize in conv_bwd.hip
    static const char* skip = FD_TUNE_GETENV("FDGAN_DEBUG_SKIP_FINALIZE");
    static long long calls = 0;
    if (skip && ++calls > atoll(skip)) return FD_OK;
  }
  BnFinArgs a{partial, rows, cpad, channels, 1.0 / (double)count, mean, var};
  return fd_launch(&bn_finalize_kernel, "bn_finalize", dim3((unsigned)((channels + 31) / 32)), dim3(1024), 0, a,
                   static_cast<hipStream_t>(stream));
}

// ---------------------------------------------------------------------------------
// NCHW fp32 -> NHWC fp16 / bf16 (zero-padded channels); one thread per (pixel, 8-channel group)
// ---------------------------------------------------------------------------------
struct ToNhwcArgs {
  const float* x;
  unsigned short* y;
  long long n, c, h, w, y_sn, y_sh, y_sw;
  int groups;  // 8-channel groups to write per pixel
  int dtype;
  long long total;
};

__global__ void nchw_to_nhwc_kernel(ToNhwcArgs a) {
  const unsigned u = blockIdx.x * blockDim.x + threadIdx.x;        // total < 2^31 (launcher): 32-bit index arithmetic
  if (u >= (unsigned)a.total) return;
  // pixel-fastest so the fp32 plane reads are coalesced
  unsigned r = u;
  const long long px = r % (unsigned)a.w;
  r /= (unsigned)a.w;
  const long long py = r % (unsigned)a.h;
  r /= (unsigned)a.h;
  const int g = (int)(r % (unsigned)a.groups);
  const long long n = r / (unsigned)a.groups;
  f32x8 v;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const long long c = g * 8 + e;
    v[e] = c < a.c ? a.x[((n * a.c + c) * a.h + py) * a.w + px] : 0.f;
  }
  *reinterpret_cast<u32x4*>(a.y + n * a.y_sn + py * a.y_sh + px * a.y_sw + g * 8) = a.dtype == FD_F16 ? fd_pk8<FmtA>(v) : fd_pk8<FmtG>(v);
}

extern "C" int fdgan_nchw_f32_to_nhwc(const float* x, int64_t n, int64_t c, int64_t h, int64_t w,
                                      const FdTensor* y, FdStream stream) {
  FD_REQUIRE(x && y && y->ptr, "nchw_f32_to_nhwc: NULL pointer");
  FD_REQUIRE((y->dtype == FD_F16 || y->dtype == FD_BF16) && y->stride[3] == 1, "nchw_f32_to_nhwc: y must be an NHWC fp16 / bf16 view");
  FD_REQUIRE(y->n == n && y->h == h && y->w == w && y->c >= c && y->c % 8 == 0,
             "nchw_f32_to_nhwc: y shape mismatch (y->c must be a multiple of 8 >= c)");
  FD_REQUIRE(y->stride[2] % 8 == 0 && y->stride[1] % 8 == 0 && y->stride[0] % 8 == 0 && ((uintptr_t)y->ptr & 15) == 0,
             "nchw_f32_to_nhwc: y alignment");
  ToNhwcArgs a;
  a.x = x;
  a.y = static_cast<unsigned short*>(y->ptr);
  a.n = n;
  a.c = c;
  a.h = h;
  a.w = w;
  a.y_sn = y->stride[0];
  a.y_sh = y->stride[1];
  a.y_sw = y->stride[2];
  a.groups = (int)(y->c / 8);
  a.dtype = y->dtype;
  a.total = n * a.groups * h * w;
  FD_REQUIRE(a.total < (1ll << 31), "nchw_f32_to_nhwc: more than 2^31 pieces");
  return fd_launch(&nchw_to_nhwc_kernel, "nchw_f32_to_nhwc", dim3((unsigned)((a.total + 255) / 256)), dim3(256),
                   0, a, static_cast<hipStream_t>(stream));
}

// ---------------------------------------------------------------------------------
// NHWC fp16 / bf16 -> NCHW fp32; one thread per (n, c, pixel): coalesced fp32 writes
// ---------------------------------------------------------------------------------
struct ToNchwArgs {
  const unsigned short* x;
  float* y;
  long long n, c, h, w, x_sn, x_sh, x_sw;
  long long total;
  int dtype;
};

__global__ void nhwc_to_nchw_kernel(ToNchwArgs a) {
  const unsigned u = blockIdx.x * blockDim.x + threadIdx.x;        // total < 2^31 (launcher): 32-bit index arithmetic
  if (u >= (unsigned)a.total) return;
  unsigned r = u;
  const long long px = r % (unsigned)a.w;
  r /= (unsigned)a.w;
  const long long py = r % (unsigned)a.h;
  r /= (unsigned)a.h;
  const long long c = r % (unsigned)a.c;
  const long long n = r / (unsigned)a.c;
  const unsigned short b = a.x[n * a.x_sn + py * a.x_sh + px * a.x_sw + c];
  a.y[u] = a.dtype == FD_F16 ? fd_cvt1<FmtA>(b) : fd_cvt1<FmtG>(b);
}

extern "C" int fdgan_nhwc_to_nchw_f32(const FdTensor* x, float* y, FdStream stream) {
  FD_REQUIRE(x && x->ptr && y, "nhwc_to_nchw_f32: NULL pointer");
  FD_REQUIRE((x->dtype == FD_F16 || x->dtype == FD_BF16) && x->stride[3] == 1, "nhwc_to_nchw_f32: x must be an NHWC fp16 / bf16 view");
  ToNchwArgs a{static_cast<const unsigned short*>(x->ptr), y, x->n, x->c, x->h, x->w,
               x->stride[0], x->stride[1], x->stride[2], x->n * x->c * x->h * x->w, x->dtype};
  FD_REQUIRE(a.total < (1ll << 31), "nhwc_to_nchw_f32: more than 2^31 elements");
  return fd_launch(&nhwc_to_nchw_kernel, "nhwc_to_nchw_f32", dim3((unsigned)((a.total + 255) / 256)), dim3(256),
                   0, a, static_cast<hipStream_t>(stream));
}

// ---------------------------------------------------------------------------------
// channel-slice copy between NHWC 16-bit views (16 B per thread)
// ---------------------------------------------------------------------------------
struct CopyArgs {
  const unsigned short* s;
  unsigned short* d;
  long long h, w, s_sn, s_sh, s_sw, d_sn, d_sh, d_sw;
  int groups;
  long long total;
};

__global__ void copy_nhwc_kernel(CopyArgs a) {
  const long long u = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= a.total) return;
  long long r = u;
  const int g = (int)(r % a.groups);
  r /= a.groups;
  const long long px = r % a.w;
  r /= a.w;
  const long long py = r % a.h;
  const long long n = r / a.h;
  const u32x4 v = *reinterpret_cast<const u32x4*>(a.s + n * a.s_sn + py * a.s_sh + px * a.s_sw + g * 8);
  *reinterpret_cast<u32x4*>(a.d + n * a.d_sn + py * a.d_sh + px * a.d_sw + g * 8) = v;
}

extern "C" int fdgan_copy_nhwc(const FdTensor* src, const FdTensor* dst, FdStream stream) {
  FD_REQUIRE(src && dst && src->ptr && dst->ptr, "copy_nhwc: NULL pointer");
  FD_REQUIRE((src->dtype == FD_F16 || src->dtype == FD_BF16) && dst->dtype == src->dtype && src->stride[3] == 1 && dst->stride[3] == 1,
             "copy_nhwc: two NHWC views of the same 16-bit format required");
  FD_REQUIRE(src->n == dst->n && src->h == dst->h && src->w == dst->w && src->c == dst->c && src->c % 8 == 0,
             "copy_nhwc: shape mismatch or c not a multiple of 8");
  FD_REQUIRE((((uintptr_t)src->ptr | (uintptr_t)dst->ptr) & 15) == 0, "copy_nhwc: 16-byte alignment");
  for (int i = 0; i < 3; ++i)
    FD_REQUIRE(src->stride[i] % 8 == 0 && dst->stride[i] % 8 == 0, "copy_nhwc: strides must be multiples of 8");
  CopyArgs a{static_cast<const unsigned short*>(src->ptr), static_cast<unsigned short*>(dst->ptr), src->h, src->w,
             src->stride[0], src->stride[1], src->stride[2], dst->stride[0], dst->stride[1], dst->stride[2],
             (int)(src->c / 8), src->n * src->h * src->w * (src->c / 8)};
  return fd_launch(&copy_nhwc_kernel, "copy_nhwc", dim3((unsigned)((a.total + 255) / 256)), dim3(256), 0, a,
                   static_cast<hipStream_t>(stream));
}

// ---------------------------------------------------------------------------------
// tanh / sigmoid in place over what a conv stored (nn.Tanh dehaze1113.py:799, nn.Sigmoid :223).
// Generic strided view (NCHW fp32 or NHWC fp16 / bf16): one thread per element, w fastest.
// ---------------------------------------------------------------------------------
struct ActArgs {
  void* y;
  long long n, c, h, w, sn, sc, sh, sw;
  int dtype, act;
  long long total;
};

__global__ void act_inplace_kernel(ActArgs a) {
  const long long u = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= a.total) return;
  long long r = u;
  long long i0, i1, i2, i3;  // fastest index first
  if (a.dtype == FD_F32) {   // NCHW: w, h, c, n
    i0 = r % a.w; r /= a.w; i1 = r % a.h; r /= a.h; i2 = r % a.c; i3 = r / a.c;
    float* p = static_cast<float*>(a.y) + i3 * a.sn + i2 * a.sc + i1 * a.sh + i0 * a.sw;
    const float v = *p;
    *p = a.act == FD_ACT_TANH ? tanhf(v) : 1.f / (1.f + expf(-v));
  } else {                   // NHWC: c, w, h, n
    i0 = r % a.c; r /= a.c; i1 = r % a.w; r /= a.w; i2 = r % a.h; i3 = r / a.h;
    unsigned short* p = static_cast<unsigned short*>(a.y) + i3 * a.sn + i2 * a.sh + i1 * a.sw + i0 * a.sc;
    const float v = a.dtype == FD_F16 ? fd_cvt1<FmtA>(*p) : fd_cvt1<FmtG>(*p);
    const float o = a.act == FD_ACT_TANH ? tanhf(v) : 1.f / (1.f + expf(-v));
    *p = a.dtype == FD_F16 ? fd_pk1<FmtA>(o) : fd_pk1<FmtG>(o);
  }
}

int fd_act_inplace(const FdTensor* y, int act, hipStream_t stream) {
  FD_REQUIRE(y && y->ptr, "act_inplace: NULL tensor");
  ActArgs a;
  a.y = y->ptr;
  a.n = y->n;
  a.c = y->c;
  a.h = y->h;
  a.w = y->w;
  a.sn = y->stride[0];
  a.sh = y->stride[1];
  a.sw = y->stride[2];
  a.sc = y->stride[3];
  a.dtype = y->dtype;
  a.act = act;
  a.total = y->n * y->c * y->h * y->w;
  return fd_launch(&act_inplace_kernel, act == FD_ACT_TANH ? "tanh_inplace" : "sigmoid_inplace",
                   dim3((unsigned)((a.total + 255) / 256)), dim3(256), 0, a, stream);
}

// ---------------------------------------------------------------------------------
// Adam (torch.optim.Adam semantics, no weight decay / amsgrad: what demo.py's lrG / lrD / beta1 flags
// configure): p -= lr * (m / (1 - b1^t)) / (sqrt(v / (1 - b2^t)) + eps) on one flat fp32 tensor.
// ---------------------------------------------------------------------------------
struct AdamArgs {
  float *p, *m, *v;
  const float* g;
  long long n;
  float lr, b1, b2, eps, bc1, bc2;   // bc = 1 - beta^t
};
__global__ __launch_bounds__(256) void adam_kernel(AdamArgs a) {
  const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= a.n) return;
  if (i + 4 <= a.n) {
    f32x4 p = *reinterpret_cast<f32x4*>(a.p + i), m = *reinterpret_cast<f32x4*>(a.m + i), v = *reinterpret_cast<f32x4*>(a.v + i);
    const f32x4 g = *reinterpret_cast<const f32x4*>(a.g + i);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      m[e] = a.b1 * m[e] + (1.f - a.b1) * g[e];
      v[e] = a.b2 * v[e] + (1.f - a.b2) * g[e] * g[e];
      p[e] -= a.lr * (m[e] / a.bc1) / (sqrtf(v[e] / a.bc2) + a.eps);
    }
    *reinterpret_cast<f32x4*>(a.p + i) = p;
    *reinterpret_cast<f32x4*>(a.m + i) = m;
    *reinterpret_cast<f32x4*>(a.v + i) = v;
  } else {
    for (long long j = i; j < a.n; ++j) {
      const float g = a.g[j];
      a.m[j] = a.b1 * a.m[j] + (1.f - a.b1) * g;
      a.v[j] = a.b2 * a.v[j] + (1.f - a.b2) * g * g;
      a.p[j] -= a.lr * (a.m[j] / a.bc1) / (sqrtf(a.v[j] / a.bc2) + a.eps);
    }
  }
}

extern "C" int fdgan_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                               float eps, int64_t step, FdStream stream) {
  FD_REQUIRE(p && g && m && v && n > 0 && step >= 1, "adam_step: bad arguments");
  FD_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "adam_step: 16-byte aligned flat tensors");
  AdamArgs a{p, m, v, g, n, lr, beta1, beta2, eps, (float)(1.0 - pow((double)beta1, (double)step)),
             (float)(1.0 - pow((double)beta2, (double)step))};
  return fd_launch(&adam_kernel, "adam_step", dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, a, static_cast<hipStream_t>(stream));
}

// ---- zero fills and the transposed accumulate of the reverse walk --------------------------------------------------
// The reverse walk of fdgan_hip/backward.py is RECORDED into an FdPlan and replayed (round 4); whatever it did with torch
// tensor methods between two launches (`buf.zero_()`, `coef[:, lo:hi].zero_()`, `grad.add_(tmp.permute(1, 0, 2, 3))`) has to
// be a launch of this library to be part of the recording.
namespace {
struct FillArgs {
  char* p;
  long long row_bytes, row_stride, rows;
};
// one thread: 64 bytes of one row (16-byte stores when the row allows it)
__global__ __launch_bounds__(256) void fill_zero_kernel(FillArgs a) {
  const long long per_row = (a.row_bytes + 63) / 64;
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= per_row * a.rows) return;
  const long long r = t / per_row, o = (t - r * per_row) * 64;
  char* d = a.p + r * a.row_stride + o;
  const long long left = a.row_bytes - o;
  if (left >= 64 && (((uintptr_t)d) & 15) == 0) {
    const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int k = 0; k < 4; ++k) reinterpret_cast<u32x4*>(d)[k] = z;
  } else {
    for (long long k = 0; k < (left < 64 ? left : 64); k += 4) *reinterpret_cast<unsigned*>(d + k) = 0u;
  }
}

struct FillManyArgs {
  const FdZeroJob* jobs;
  long long njobs;
};
// one workgroup: 16 KiB of one job (binary search over the jobs' first groups)
__global__ __launch_bounds__(256) void fill_zero_many_kernel(FillManyArgs a) {
  const long long grp = blockIdx.x;
  long long lo = 0, hi = a.njobs - 1;
  while (lo < hi) {
    const long long mid = (lo + hi + 1) >> 1;
    if (a.jobs[mid].first_group <= grp) lo = mid; else hi = mid - 1;
  }
  const FdZeroJob j = a.jobs[lo];
  const long long o = (grp - j.first_group) * 16384 + (long long)threadIdx.x * 16;
  char* d = static_cast<char*>(j.ptr) + o;
  const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (o + k * 4096 + 16 <= j.bytes) *reinterpret_cast<u32x4*>(d + k * 4096) = z;
}

struct AddTArgs {
  float* dst;
  const float* src;
  long long rows, cols;
};
// dst[c][r] += src[r][c]   (dst: cols x rows)
__global__ __launch_bounds__(256) void add_transposed_kernel(AddTArgs a) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= a.rows * a.cols) return;
  const long long c = t / a.rows, r = t - c * a.rows;      // consecutive threads: consecutive dst elements
  a.dst[t] += a.src[r * a.cols + c];
}
}  // namespace

extern "C" int fdgan_fill_zero(void* p, int64_t row_bytes, int64_t rows, int64_t row_stride_bytes, FdStream stream) {
  FD_REQUIRE(p && row_bytes > 0 && rows > 0 && (row_bytes & 3) == 0 && (((uintptr_t)p) & 3) == 0 && (row_stride_bytes & 3) == 0,
             "fill_zero: %lld rows of %lld bytes (4-byte granularity)", (long long)rows, (long long)row_bytes);
  FillArgs a{static_cast<char*>(p), row_bytes, row_stride_bytes, rows};
  const long long threads = (row_bytes + 63) / 64 * rows;
  return fd_launch(&fill_zero_kernel, "fill_zero", dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, a, static_cast<hipStream_t>(stream));
}

extern "C" int fdgan_fill_zero_many(const FdZeroJob* jobs_device, int64_t njobs, int64_t total_groups, FdStream stream) {
  FD_REQUIRE(jobs_device && njobs > 0 && total_groups > 0 && total_groups < (1ll << 31), "fill_zero_many: %lld jobs, %lld groups",
             (long long)njobs, (long long)total_groups);
  FillManyArgs a{jobs_device, njobs};
  return fd_launch(&fill_zero_many_kernel, "fill_zero_many", dim3((unsigned)total_groups), dim3(256), 0, a, static_cast<hipStream_t>(stream));
}

extern "C" int fdgan_add_transposed_f32(float* dst, const float* src, int64_t rows, int64_t cols, FdStream stream) {
  FD_REQUIRE(dst && src && rows > 0 && cols > 0, "add_transposed_f32: bad arguments");
  AddTArgs a{dst, src, rows, cols};
  return fd_launch(&add_transposed_kernel, "add_transposed_f32", dim3((unsigned)((rows * cols + 255) / 256)), dim3(256), 0, a,
                   static_cast<hipStream_t>(stream));
}

// ---- element-wise dropout of the dy blocks (dropRate > 0; /root/reference/models/dehaze1113.py:270-274, :367-368) --------
// F.dropout(out, p, training) = out * mask / (1 - p) with an element-wise Bernoulli mask.  The host draws the mask (torch's
// generator, as the reference does) into an NHWC fp16 tensor that already holds 0 or 1 / (1 - p); forward and backward are the same
// in-place multiply, on an fp16 activation view or a bf16 gradient view.  up2: the view is the 2x nearest-upsampled image of
// what the mask covers (TransitionBlockdy drops BEFORE it upsamples): pixel (y, x) takes mask (y / 2, x / 2).
namespace {
struct MulMaskArgs {
  unsigned short* d;
  long long d_sn;
  int d_sh, d_sw;
  const unsigned short* m;
  long long m_sn;
  int m_sh, m_sw;
  int N, H, W, C8, up2, grad;
};
__global__ __launch_bounds__(256) void mul_mask_kernel(MulMaskArgs a) {
  const unsigned u = blockIdx.x * 256u + threadIdx.x;
  if (u >= (unsigned)a.N * a.H * a.W * a.C8) return;
  const int c8 = (int)(u % (unsigned)a.C8);
  unsigned r = u / (unsigned)a.C8;
  const int x = (int)(r % (unsigned)a.W);
  r /= (unsigned)a.W;
  const int y = (int)(r % (unsigned)a.H), n = (int)(r / (unsigned)a.H);
  unsigned short* dp = a.d + n * a.d_sn + (long long)y * a.d_sh + (long long)x * a.d_sw + c8 * 8;
  const unsigned short* mp = a.m + n * a.m_sn + (long long)(a.up2 ? y >> 1 : y) * a.m_sh + (long long)(a.up2 ? x >> 1 : x) * a.m_sw + c8 * 8;
  const f32x8 mk = fd_cvt8<FmtA>(*reinterpret_cast<const u32x4*>(mp));
  const u32x4 raw = *reinterpret_cast<const u32x4*>(dp);
  if (a.grad) *reinterpret_cast<u32x4*>(dp) = fd_pk8<FmtG>(fd_cvt8<FmtG>(raw) * mk);
  else *reinterpret_cast<u32x4*>(dp) = fd_pk8<FmtA>(fd_cvt8<FmtA>(raw) * mk);
}
}  // namespace

extern "C" int fdgan_mul_mask_nhwc(const FdTensor* mask, const FdTensor* dst, int up2, FdStream stream) {
  FD_REQUIRE(mask && dst && mask->ptr && dst->ptr, "mul_mask_nhwc: NULL tensor");
  FD_REQUIRE(mask->dtype == FD_F16 && (dst->dtype == FD_F16 || dst->dtype == FD_BF16), "mul_mask_nhwc: an fp16 mask and a 16-bit view");
  const int f = up2 ? 2 : 1;
  // whole 8-channel groups only: the tail group of a view in the middle of a concat buffer would cover (and multiply by the mask's
  // zero padding) the NEXT slice's first channels
  FD_REQUIRE(dst->c % 8 == 0, "mul_mask_nhwc: %lld channels (a multiple of 8 expected)", (long long)dst->c);
  FD_REQUIRE(mask->n == dst->n && mask->h * f == dst->h && mask->w * f == dst->w && mask->c == dst->c, "mul_mask_nhwc: shape mismatch");
  for (const FdTensor* t : {mask, dst})
    FD_REQUIRE(t->stride[3] == 1 && t->stride[2] % 8 == 0 && t->stride[1] % 8 == 0 && t->stride[0] % 8 == 0 && ((uintptr_t)t->ptr & 15) == 0,
               "mul_mask_nhwc: NHWC 16-bit views with 8-element aligned strides expected");
  MulMaskArgs a{static_cast<unsigned short*>(dst->ptr), dst->stride[0], (int)dst->stride[1], (int)dst->stride[2],
                static_cast<const unsigned short*>(mask->ptr), mask->stride[0], (int)mask->stride[1], (int)mask->stride[2],
                (int)dst->n, (int)dst->h, (int)dst->w, (int)((dst->c + 7) / 8), up2 ? 1 : 0, dst->dtype == FD_BF16 ? 1 : 0};
  const long long total = (long long)a.N * a.H * a.W * a.C8;
  FD_REQUIRE(total > 0 && total < (1ll << 31), "mul_mask_nhwc: %lld groups", total);
  return fd_launch(&mul_mask_kernel, "mul_mask_nhwc", dim3((unsigned)((total + 255) / 256)), dim3(256), 0, a, static_cast<hipStream_t>(stream));
}
