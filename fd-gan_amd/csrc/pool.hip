// pool.hip -- 2x2 stride-2 max pooling on NHWC fp16 views (F.max_pool2d(h, 2, 2),
// /root/reference/myutils/vgg16.py:31,36,42).  HBM-bound: 16 bytes (8 channels) per thread.
#include "common.h"

namespace {
struct PoolArgs {
  const unsigned short* x;
  unsigned short* y;
  long long x_sn, x_sh, x_sw, y_sn, y_sh, y_sw;
  int Ho, Wo, groups;
  long long total;
};

__global__ void maxpool2_kernel(PoolArgs a) {
  const unsigned u = blockIdx.x * blockDim.x + threadIdx.x;      // total < 2^31 (launcher): 32-bit index arithmetic (a 64-bit
  if (u >= (unsigned)a.total) return;                             // division is ~100 vector instructions; there were four per thread)
  unsigned r = u;
  const int g = (int)(r % (unsigned)a.groups);
  r /= (unsigned)a.groups;
  const int ox = (int)(r % (unsigned)a.Wo);
  r /= (unsigned)a.Wo;
  const int oy = (int)(r % (unsigned)a.Ho);
  const long long n = r / (unsigned)a.Ho;
  const unsigned short* p = a.x + n * a.x_sn + (long long)(2 * oy) * a.x_sh + (long long)(2 * ox) * a.x_sw + g * 8;
  const f32x8 f0 = fd_cvt8<FmtA>(*reinterpret_cast<const u32x4*>(p)), f1 = fd_cvt8<FmtA>(*reinterpret_cast<const u32x4*>(p + a.x_sw));
  const f32x8 f2 = fd_cvt8<FmtA>(*reinterpret_cast<const u32x4*>(p + a.x_sh)), f3 = fd_cvt8<FmtA>(*reinterpret_cast<const u32x4*>(p + a.x_sh + a.x_sw));
  f32x8 m;
#pragma unroll
  for (int e = 0; e < 8; ++e) m[e] = fmaxf(fmaxf(f0[e], f1[e]), fmaxf(f2[e], f3[e]));
  *reinterpret_cast<u32x4*>(a.y + n * a.y_sn + (long long)oy * a.y_sh + (long long)ox * a.y_sw + g * 8) = fd_pk8<FmtA>(m);
}
// backward: every input position belongs to exactly one 2x2 window; the gradient of the window goes to its
// FIRST maximum in row-major order (what F.max_pool2d's backward does on ties -- frequent after a ReLU), and is
// ADDED to dx (the pooled tensor's source is usually also a tapped feature map with its own gradient).
struct PoolBwdArgs {
  const unsigned short *x, *dy;
  unsigned short* dx;
  long long x_sn, x_sh, x_sw, dy_sn, dy_sh, dy_sw, dx_sn, dx_sh, dx_sw;
  int Ho, Wo, groups;
  long long total;
};
__global__ void maxpool2_bwd_kernel(PoolBwdArgs a) {
  const unsigned u = blockIdx.x * blockDim.x + threadIdx.x;      // total < 2^31 (launcher): 32-bit index arithmetic (a 64-bit
  if (u >= (unsigned)a.total) return;                             // division is ~100 vector instructions; there were four per thread)
  unsigned r = u;
  const int g = (int)(r % (unsigned)a.groups);
  r /= (unsigned)a.groups;
  const int ox = (int)(r % (unsigned)a.Wo);
  r /= (unsigned)a.Wo;
  const int oy = (int)(r % (unsigned)a.Ho);
  const long long n = r / (unsigned)a.Ho;
  const unsigned short* p = a.x + n * a.x_sn + (long long)(2 * oy) * a.x_sh + (long long)(2 * ox) * a.x_sw + g * 8;
  const long long xo[4] = {0, a.x_sw, a.x_sh, a.x_sh + a.x_sw};
  f32x8 f[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) f[q] = fd_cvt8<FmtA>(*reinterpret_cast<const u32x4*>(p + xo[q]));   // x: the forward activation (fp16); dy / dx: bf16
  const f32x8 d = __builtin_convertvector(
      __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(a.dy + n * a.dy_sn + (long long)oy * a.dy_sh + (long long)ox * a.dy_sw + g * 8)), f32x8);
  int arg[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    int am = 0;
    float mv = f[0][e];
#pragma unroll
    for (int q = 1; q < 4; ++q)
      if (f[q][e] > mv) {
        mv = f[q][e];
        am = q;
      }
    arg[e] = am;
  }
  unsigned short* o = a.dx + n * a.dx_sn + (long long)(2 * oy) * a.dx_sh + (long long)(2 * ox) * a.dx_sw + g * 8;
  const long long dxo[4] = {0, a.dx_sw, a.dx_sh, a.dx_sh + a.dx_sw};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    f32x8 cur = __builtin_convertvector(__builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(o + dxo[q])), f32x8);
#pragma unroll
    for (int e = 0; e < 8; ++e) cur[e] += arg[e] == q ? d[e] : 0.f;
    *reinterpret_cast<u32x4*>(o + dxo[q]) = __builtin_bit_cast(u32x4, __builtin_convertvector(cur, bf16x8));
  }
}
}  // namespace

extern "C" int fdgan_maxpool2_bwd_nhwc(const FdTensor* x, const FdTensor* dy, const FdTensor* dx, FdStream stream) {
  FD_REQUIRE(x && dy && dx && x->ptr && dy->ptr && dx->ptr, "maxpool2_bwd_nhwc: NULL pointer");
  FD_REQUIRE(x->dtype == FD_F16 && dy->dtype == FD_BF16 && dx->dtype == FD_BF16 && x->stride[3] == 1 && dy->stride[3] == 1 &&
                 dx->stride[3] == 1,
             "maxpool2_bwd_nhwc: x an NHWC fp16 activation, dy / dx NHWC bf16 gradients");
  FD_REQUIRE(dy->n == x->n && dy->h == x->h / 2 && dy->w == x->w / 2 && dy->c == x->c && x->c % 8 == 0 && dx->n == x->n &&
                 dx->h == x->h && dx->w == x->w && dx->c == x->c,
             "maxpool2_bwd_nhwc: shape mismatch (c must be a multiple of 8)");
  FD_REQUIRE((((uintptr_t)x->ptr | (uintptr_t)dy->ptr | (uintptr_t)dx->ptr) & 15) == 0, "maxpool2_bwd_nhwc: 16-byte alignment");
  for (int i = 0; i < 3; ++i)
    FD_REQUIRE(x->stride[i] % 8 == 0 && dy->stride[i] % 8 == 0 && dx->stride[i] % 8 == 0, "maxpool2_bwd_nhwc: strides must be multiples of 8");
  PoolBwdArgs a{static_cast<const unsigned short*>(x->ptr), static_cast<const unsigned short*>(dy->ptr),
                static_cast<unsigned short*>(dx->ptr), x->stride[0], x->stride[1], x->stride[2], dy->stride[0], dy->stride[1],
                dy->stride[2], dx->stride[0], dx->stride[1], dx->stride[2], (int)dy->h, (int)dy->w, (int)(x->c / 8),
                dy->n * dy->h * dy->w * (x->c / 8)};
  FD_REQUIRE(a.total > 0 && a.total < (1ll << 31), "maxpool2_bwd_nhwc: empty, or more than 2^31 pieces");
  return fd_launch(&maxpool2_bwd_kernel, "maxpool2_bwd_nhwc", dim3((unsigned)((a.total + 255) / 256)), dim3(256), 0, a,
                   static_cast<hipStream_t>(stream));
}

extern "C" int fdgan_maxpool2_nhwc(const FdTensor* x, const FdTensor* y, FdStream stream) {
  FD_REQUIRE(x && y && x->ptr && y->ptr, "maxpool2_nhwc: NULL pointer");
  FD_REQUIRE(x->dtype == FD_F16 && y->dtype == FD_F16 && x->stride[3] == 1 && y->stride[3] == 1,
             "maxpool2_nhwc: NHWC fp16 views required");
  FD_REQUIRE(y->n == x->n && y->h == x->h / 2 && y->w == x->w / 2 && y->c == x->c && x->c % 8 == 0,
             "maxpool2_nhwc: shape mismatch (c must be a multiple of 8)");
  FD_REQUIRE((((uintptr_t)x->ptr | (uintptr_t)y->ptr) & 15) == 0, "maxpool2_nhwc: 16-byte alignment");
  for (int i = 0; i < 3; ++i)
    FD_REQUIRE(x->stride[i] % 8 == 0 && y->stride[i] % 8 == 0, "maxpool2_nhwc: strides must be multiples of 8");
  PoolArgs a{static_cast<const unsigned short*>(x->ptr), static_cast<unsigned short*>(y->ptr),
             x->stride[0], x->stride[1], x->stride[2], y->stride[0], y->stride[1], y->stride[2],
             (int)y->h, (int)y->w, (int)(x->c / 8), y->n * y->h * y->w * (x->c / 8)};
  FD_REQUIRE(a.total > 0 && a.total < (1ll << 31), "maxpool2_nhwc: empty output, or more than 2^31 pieces");
  return fd_launch(&maxpool2_kernel, "maxpool2_nhwc", dim3((unsigned)((a.total + 255) / 256)), dim3(256), 0, a,
                   static_cast<hipStream_t>(stream));
}
