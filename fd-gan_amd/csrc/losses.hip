// losses.hip -- the scalar reductions between the networks' outputs and the number that drives backward:
// L1 / MSE / BCE means on fp32 tensors (image-sized: 3 x 256 x 256, D's 126 x 126 sigmoid map) with their gradient in
// the same pass, and the perceptual MSE between two NHWC fp16 feature maps of Vgg16 read in place -- no NCHW fp32
// copies of the feature maps, no elementwise torch kernels.  Reductions are two-stage and ordered (per-workgroup
// partials, then one workgroup sums them in index order in fp64): bit-reproducible run to run.
//
// Reference semantics: torch.nn.functional.l1_loss / mse_loss / binary_cross_entropy as a training loop over
// /root/reference/models/dehaze1113.py:188-230 (D's sigmoid map) and myutils/vgg16.py:27-49 (feature maps) composes them.
#include "common.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}

// block of 256 threads -> one float in partial[blockIdx.x]
__device__ __forceinline__ void block_store_sum(float v, float* partial) {
  __shared__ float red[4];
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

struct LossArgs {
  const float* x;
  const float* t;
  float t_const;
  long long n;
  float inv_n;
  float* grad;
  float* partial;
  int kind;
};

__global__ __launch_bounds__(256) void loss_f32_kernel(LossArgs a) {
  float acc = 0.f;
  const long long n4 = a.n >> 2;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4 + 1; i += (long long)gridDim.x * 256) {
    float xv[4], tv[4], gv[4];
    const bool full = i < n4;
    const long long base = i * 4;
    int cnt = 4;
    if (full) {
      const f32x4 x4 = *reinterpret_cast<const f32x4*>(a.x + base);
      xv[0] = x4[0], xv[1] = x4[1], xv[2] = x4[2], xv[3] = x4[3];
      if (a.t) {
        const f32x4 t4 = *reinterpret_cast<const f32x4*>(a.t + base);
        tv[0] = t4[0], tv[1] = t4[1], tv[2] = t4[2], tv[3] = t4[3];
      }
    } else {   // ragged tail
      cnt = (int)(a.n - base);
      for (int e = 0; e < cnt; ++e) {
        xv[e] = a.x[base + e];
        if (a.t) tv[e] = a.t[base + e];
      }
    }
    if (!a.t)
      for (int e = 0; e < 4; ++e) tv[e] = a.t_const;
    for (int e = 0; e < cnt; ++e) {
      const float d = xv[e] - tv[e];
      float l, g;
      if (a.kind == 0) {          // L1; sign(0) = 0 as torch
        l = fabsf(d);
        g = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
      } else if (a.kind == 1) {   // MSE
        l = d * d;
        g = 2.f * d;
      } else {                    // BCE, logs clamped at -100 (torch.nn.functional.binary_cross_entropy)
        const float lx = fmaxf(logf(xv[e]), -100.f), l1x = fmaxf(logf(1.f - xv[e]), -100.f);
        l = -(tv[e] * lx + (1.f - tv[e]) * l1x);
        g = d / fmaxf((1.f - xv[e]) * xv[e], 1e-12f);
      }
      acc += l;
      gv[e] = g * a.inv_n;
    }
    if (a.grad) {
      if (full) *reinterpret_cast<f32x4*>(a.grad + base) = f32x4{gv[0], gv[1], gv[2], gv[3]};
      else
        for (int e = 0; e < cnt; ++e) a.grad[base + e] = gv[e];
    }
  }
  block_store_sum(acc, a.partial);
}

struct SumArgs {
  const float* partial;
  long long count;
  double scale;
  float* out;
};

// one workgroup; thread t sums partial[t], partial[t + 256], ... then the 256 sums are added in index order (fp64)
__global__ __launch_bounds__(256) void sum_partials_kernel(SumArgs a) {
  __shared__ double sh[256];
  double s = 0.0;
  for (long long i = threadIdx.x; i < a.count; i += 256) s += (double)a.partial[i];
  sh[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < 256; ++i) t += sh[i];
    a.out[0] = (float)(t * a.scale);
  }
}

struct MseNhwcArgs {
  const unsigned short *a, *b;
  long long a_sn, b_sn, g_sn;
  int a_sh, a_sw, b_sh, b_sw, g_sh, g_sw;
  int H, W, C8;              // C8 = channels / 8
  long long units;           // N * H * W * C8 pieces of 8 channels
  float scale;
  const float* upstream;     // backward: device scalar (d total / d this loss)
  int relu_mask;             // backward: g = 0 where a == 0 (a is a ReLU output: its pre-activation was <= 0 there)
  unsigned short* g;         // backward: gradient view (bf16)
  float* partial;            // forward
  int flat;                  // every view is a whole dense buffer: mse_nhwc_flat_kernel
};

// FLAT: every view is a whole dense buffer (VGG16's taps are) -- piece u sits at element 8 u, no index arithmetic, and four pieces
// per operand are requested before the first is used (round 6: the generic form has three integer divisions and ONE load pair per
// thread in flight: 107 us for 16 x 64 x 256^2 = 2.5 TB/s).  Same thread -> piece assignment and the same order of additions as
// the generic form: bitwise the same sums.
template <bool BWD>
__global__ __launch_bounds__(256) void mse_nhwc_flat_kernel(MseNhwcArgs a) {
  float acc = 0.f;
  float up = 0.f;
  if (BWD) up = a.upstream[0] * a.scale;
  const unsigned units = (unsigned)a.units, step = gridDim.x * 256u;
  const u32x4* ap = reinterpret_cast<const u32x4*>(a.a);
  const u32x4* bp = reinterpret_cast<const u32x4*>(a.b);
  u32x4* gp = reinterpret_cast<u32x4*>(a.g);
  auto one = [&](unsigned u, const u32x4 av, const u32x4 bv) {
    const f32x8 d = fd_cvt8<FmtA>(av) - fd_cvt8<FmtA>(bv);
    if (BWD) {
      f32x8 gv = d * up;
      if (a.relu_mask) {
        const f32x8 af = fd_cvt8<FmtA>(av);
#pragma unroll
        for (int e = 0; e < 8; ++e) gv[e] = af[e] > 0.f ? gv[e] : 0.f;
      }
      gp[u] = __builtin_bit_cast(u32x4, __builtin_convertvector(gv, bf16x8));
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) acc += d[e] * d[e];
    }
  };
  unsigned u = blockIdx.x * 256u + threadIdx.x;
  for (; u < units && units - u > 3u * step; u += 4u * step) {
    u32x4 av[4], bv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) av[k] = ap[u + k * step], bv[k] = bp[u + k * step];
#pragma unroll
    for (int k = 0; k < 4; ++k) one(u + k * step, av[k], bv[k]);
  }
  for (; u < units; u += step) one(u, ap[u], bp[u]);
  if (!BWD) block_store_sum(acc * a.scale, a.partial);
}

template <bool BWD>
__global__ __launch_bounds__(256) void mse_nhwc_kernel(MseNhwcArgs a) {
  float acc = 0.f;
  float up = 0.f;
  if (BWD) up = a.upstream[0] * a.scale;
  for (unsigned u = blockIdx.x * 256u + threadIdx.x; u < (unsigned)a.units; u += gridDim.x * 256u) {   // units < 2^31: 32-bit index arithmetic
    const int c8 = (int)(u % (unsigned)a.C8);
    unsigned p = u / (unsigned)a.C8;
    const int x = (int)(p % (unsigned)a.W);
    p /= (unsigned)a.W;
    const int y = (int)(p % (unsigned)a.H);
    const long long n = p / (unsigned)a.H;
    const u32x4 av = *reinterpret_cast<const u32x4*>(a.a + n * a.a_sn + (long long)y * a.a_sh + (long long)x * a.a_sw + c8 * 8);
    const u32x4 bv = *reinterpret_cast<const u32x4*>(a.b + n * a.b_sn + (long long)y * a.b_sh + (long long)x * a.b_sw + c8 * 8);
    const f32x8 d = fd_cvt8<FmtA>(av) - fd_cvt8<FmtA>(bv);      // feature maps: fp16
    if (BWD) {
      f32x8 gv = d * up;
      if (a.relu_mask) {
        const f32x8 af = fd_cvt8<FmtA>(av);
#pragma unroll
        for (int e = 0; e < 8; ++e) gv[e] = af[e] > 0.f ? gv[e] : 0.f;
      }
      *reinterpret_cast<u32x4*>(a.g + n * a.g_sn + (long long)y * a.g_sh + (long long)x * a.g_sw + c8 * 8) =
          __builtin_bit_cast(u32x4, __builtin_convertvector(gv, bf16x8));
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) acc += d[e] * d[e];
    }
  }
  if (!BWD) block_store_sum(acc * a.scale, a.partial);
}

int mse_setup(const FdTensor* a, const FdTensor* b, const FdTensor* g, MseNhwcArgs& m) {
  FD_REQUIRE(a && b, "mse_nhwc: NULL tensor");
  for (const FdTensor* t : {a, b, g}) {
    if (!t) continue;
    FD_REQUIRE(t->dtype == (t == g ? FD_BF16 : FD_F16) && t->stride[3] == 1 && ((uintptr_t)t->ptr & 15) == 0 && t->stride[2] % 8 == 0 &&
                   t->stride[1] % 8 == 0 && t->stride[0] % 8 == 0,
               "mse_nhwc: NHWC fp16 feature views (bf16 gradient view) with 16-byte aligned pixels");
    FD_REQUIRE(t->n == a->n && t->h == a->h && t->w == a->w && t->c == a->c, "mse_nhwc: shapes differ");
  }
  FD_REQUIRE(a->c % 8 == 0 && a->c > 0, "mse_nhwc: channels must be a multiple of 8");
  m = MseNhwcArgs{};
  m.a = static_cast<const unsigned short*>(a->ptr);
  m.b = static_cast<const unsigned short*>(b->ptr);
  m.a_sn = a->stride[0], m.a_sh = (int)a->stride[1], m.a_sw = (int)a->stride[2];
  m.b_sn = b->stride[0], m.b_sh = (int)b->stride[1], m.b_sw = (int)b->stride[2];
  if (g) m.g = static_cast<unsigned short*>(g->ptr), m.g_sn = g->stride[0], m.g_sh = (int)g->stride[1], m.g_sw = (int)g->stride[2];
  m.H = (int)a->h, m.W = (int)a->w, m.C8 = (int)(a->c / 8);
  m.units = a->n * a->h * a->w * m.C8;
  FD_REQUIRE(m.units < (1ll << 31), "mse_nhwc: more than 2^31 pieces");
  m.flat = 1;
  for (const FdTensor* t : {a, b, g})
    if (t && !(t->stride[2] == t->c && t->stride[1] == t->w * t->c && t->stride[0] == t->h * t->w * t->c)) m.flat = 0;
  return FD_OK;
}

// ---- ContextualLoss rows (loss.py:49-68 of the reference's bytecode-only loss module) ------------------
// For one row i of the cosine-distance matrix d[b][i][:] the chain  relative_distances -> weighted_average_distances ->
// max over j  collapses to a softmax-style reduction:
//     r_ij = d_ij / (dmin_i + e),  w_ij = exp((b - r_ij) / sigma),  cx_ij = w_ij / sum_j w_ij,
//     m_i = max_j cx_ij = 1 / S_i,   S_i = sum_j exp(u_ij),   u_ij = (dmin_i - d_ij) / (sigma (dmin_i + e))   (b cancels)
// so the three HW x HW intermediates (r, w, cx) of the torch formulation never exist.  One wave per row.
// Backward, with g_i = dL/dm_i:  dL/dd_ij = g_i / S_i^2 * E_ij / (sigma (dmin_i + e))  for j != argmin, and for the argmin
// (whose own term is identically 1)  - g_i / S_i^2 * sum_{j != j*} E_ij (d_ij + e) / (sigma (dmin_i + e)^2).
struct CxArgs {
  const float* d;        // [rows][n] contiguous
  long long rows;
  int n;
  float sigma, eps;
  float* m;              // [rows]   (forward out)
  float* dmin;           // [rows]   (forward out, backward in)
  float* S;              // [rows]
  int* jmin;             // [rows]   first argmin
  const float* gm;       // backward: dL/dm
  float* gd;             // backward: dL/dd [rows][n]
};

template <bool BWD>
__global__ __launch_bounds__(256) void cx_rows_kernel(CxArgs a) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.rows) return;
  const float* dr = a.d + row * a.n;
  if (!BWD) {
    float mn = 3.4e38f;
    int jm = 0x7fffffff;
    for (int j = lane; j < a.n; j += 64) {
      const float v = dr[j];
      if (v < mn) mn = v, jm = j;       // strided ascending: first occurrence within the lane
    }
#pragma unroll
    for (int dlt = 32; dlt > 0; dlt >>= 1) {
      const float om = __shfl_xor(mn, dlt, 64);
      const int oj = __shfl_xor(jm, dlt, 64);
      if (om < mn || (om == mn && oj < jm)) mn = om, jm = oj;
    }
    if (jm < 0 || jm >= a.n) jm = 0;    // an all-NaN row (0 / 0 in the cosine of a zero-norm feature): `v < mn` never fired;
                                        // the loss is NaN as in torch, but the index the backward writes through stays in the row
    const float inv = 1.f / (a.sigma * (mn + a.eps));
    float s = 0.f;
    for (int j = lane; j < a.n; j += 64) s += __expf((mn - dr[j]) * inv);
    s = wave_sum(s);
    if (lane == 0) {
      a.m[row] = 1.f / s;
      a.dmin[row] = mn;
      a.S[row] = s;
      a.jmin[row] = jm;
    }
  } else {
    const float mn = a.dmin[row], s = a.S[row], g = a.gm[row];
    const int jm = a.jmin[row];
    const float inv = 1.f / (a.sigma * (mn + a.eps));
    const float k = g / (s * s) * inv;                 // g / S^2 / (sigma (dmin + e))
    float* gr = a.gd + row * a.n;
    float acc = 0.f;
    for (int j = lane; j < a.n; j += 64) {
      const float dv = dr[j];
      const float e = __expf((mn - dv) * inv);
      if (j != jm) {
        gr[j] = k * e;
        acc += e * (dv + a.eps);
      }
    }
    acc = wave_sum(acc);
    if (lane == 0 && jm >= 0 && jm < a.n) gr[jm] = -k * acc / (mn + a.eps);
  }
}

unsigned grid_for(long long work_items) {   // a few workgroups per CU, grid-stride beyond
  const long long nb = (work_items + 255) / 256;
  return (unsigned)(nb < 1 ? 1 : (nb > 2048 ? 2048 : nb));
}

}  // namespace

extern "C" int fdgan_loss_f32(int kind, const float* x, const float* t, float t_const, int64_t n, float* grad, float* partial,
                              int64_t partial_floats, int64_t* nparts, FdStream stream) {
  FD_REQUIRE(kind >= 0 && kind <= 2, "loss_f32: kind %d", kind);
  FD_REQUIRE(x && partial && n > 0, "loss_f32: NULL pointer / empty tensor");
  FD_REQUIRE(((uintptr_t)x & 15) == 0 && (!t || ((uintptr_t)t & 15) == 0) && (!grad || ((uintptr_t)grad & 15) == 0),
             "loss_f32: 16-byte aligned tensors");
  LossArgs a{x, t, t_const, (long long)n, 1.f / (float)n, grad, partial, kind};
  const unsigned nb = grid_for((n + 3) / 4 + 1);
  FD_REQUIRE(partial_floats >= (int64_t)nb, "loss_f32: partial workspace too small (%u needed)", nb);
  if (nparts) *nparts = nb;
  return fd_launch(&loss_f32_kernel, "loss_f32", dim3(nb), dim3(256), 0, a, static_cast<hipStream_t>(stream));
}

extern "C" int fdgan_sum_partials(const float* partial, int64_t count, double scale, float* out, FdStream stream) {
  FD_REQUIRE(partial && out && count > 0, "sum_partials: NULL pointer / empty");
  SumArgs a{partial, (long long)count, scale, out};
  return fd_launch(&sum_partials_kernel, "sum_partials", dim3(1), dim3(256), 0, a, static_cast<hipStream_t>(stream));
}

extern "C" int fdgan_mse_nhwc_fwd(const FdTensor* a, const FdTensor* b, float scale, float* partial, int64_t partial_floats,
                                  int64_t* nparts, FdStream stream) {
  MseNhwcArgs m;
  int rc = mse_setup(a, b, nullptr, m);
  if (rc != FD_OK) return rc;
  FD_REQUIRE(partial, "mse_nhwc_fwd: NULL partial");
  const unsigned nb = grid_for(m.units);
  FD_REQUIRE(partial_floats >= (int64_t)nb, "mse_nhwc_fwd: partial workspace too small (%u needed)", nb);
  m.scale = scale;
  m.partial = partial;
  if (nparts) *nparts = nb;
  if (m.flat) return fd_launch(&mse_nhwc_flat_kernel<false>, "mse_nhwc_fwd", dim3(nb), dim3(256), 0, m, static_cast<hipStream_t>(stream));
  return fd_launch(&mse_nhwc_kernel<false>, "mse_nhwc_fwd", dim3(nb), dim3(256), 0, m, static_cast<hipStream_t>(stream));
}

extern "C" int fdgan_mse_nhwc_bwd(const FdTensor* a, const FdTensor* b, const float* upstream, float scale, int relu_mask,
                                  const FdTensor* g, FdStream stream) {
  MseNhwcArgs m;
  FD_REQUIRE(g && upstream, "mse_nhwc_bwd: NULL gradient view / upstream scalar");
  int rc = mse_setup(a, b, g, m);
  if (rc != FD_OK) return rc;
  m.scale = scale;
  m.upstream = upstream;
  m.relu_mask = relu_mask;
  if (m.flat) return fd_launch(&mse_nhwc_flat_kernel<true>, "mse_nhwc_bwd", dim3(grid_for(m.units)), dim3(256), 0, m, static_cast<hipStream_t>(stream));
  return fd_launch(&mse_nhwc_kernel<true>, "mse_nhwc_bwd", dim3(grid_for(m.units)), dim3(256), 0, m, static_cast<hipStream_t>(stream));
}

extern "C" int fdgan_cx_rows_fwd(const float* d, int64_t rows, int64_t n, float sigma, float eps, float* m, float* dmin, float* S,
                                 int32_t* jmin, FdStream stream) {
  FD_REQUIRE(d && m && dmin && S && jmin && rows > 0 && n > 0 && n < (1ll << 30) && sigma > 0.f, "cx_rows_fwd: bad arguments");
  CxArgs a{d, (long long)rows, (int)n, sigma, eps, m, dmin, S, jmin, nullptr, nullptr};
  return fd_launch(&cx_rows_kernel<false>, "cx_rows_fwd", dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, a, static_cast<hipStream_t>(stream));
}

extern "C" int fdgan_cx_rows_bwd(const float* d, int64_t rows, int64_t n, float sigma, float eps, const float* dmin, const float* S,
                                 const int32_t* jmin, const float* gm, float* gd, FdStream stream) {
  FD_REQUIRE(d && dmin && S && jmin && gm && gd && rows > 0 && n > 0 && n < (1ll << 30) && sigma > 0.f, "cx_rows_bwd: bad arguments");
  CxArgs a{d, (long long)rows, (int)n, sigma, eps, nullptr, const_cast<float*>(dmin), const_cast<float*>(S), const_cast<int*>(jmin), gm, gd};
  return fd_launch(&cx_rows_kernel<true>, "cx_rows_bwd", dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, a, static_cast<hipStream_t>(stream));
}
