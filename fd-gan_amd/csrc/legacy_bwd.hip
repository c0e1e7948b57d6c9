// legacy_bwd.hip -- reverse-mode counterparts of legacy.hip: what torch.autograd does for the DCPDN-era networks
// (models/dehaze22.py G :205-362, G2 :364-488, Dense :531-660, dehaze :662-753; models/dehaze1113.py Dense :431-570, Dense2
// :572-699) between their convolutions.  None of this is on the benchmark's path: the kernels are written for clarity and a
// fixed summation order, one pass each.
//
//   fdgan_maxpool3s2_bwd     gradient of MaxPool2d(3, 2, 1)(act(bn(x))) w.r.t. the ACTIVATED tensor: every input position
//                            collects dy of the (at most four) windows whose first maximum it is -- ATen's tie rule, scan order
//                            -- ; the activation mask and BatchNorm's backward then run as for any prologue
//                            (fdgan_bn_act_bwd / fdgan_bn_bwd_finalize / fdgan_bn_bwd_apply).
//   fdgan_pyramid_pool4_bwd  the four-scale head: with coef_j[p] = leaky'(z_j[window_j(p)]) * sum_{q in window_j(p)} dy[q][j] / k_j^2,
//                            dx[p][c] += sum_j coef_j[p] w_j[c],  dw_j[c] = sum_p coef_j[p] x[p][c],  db_j = sum_windows (...);
//                            dw / db leave as one partial row per k0 x k0 tile.
//   fdgan_bn_dropout_bwd     y = mask[n][c] * bn(x) with batch statistics (the U-Net's three innermost decoder outputs, at most
//                            8 x 8 pixels): dx, dgamma, dbeta, one workgroup per channel.
//   fdgan_scatter_dehaze_bwd J = (I - A) / (|t| + eps) + A,  A = leaky(window mean of atp): gradients w.r.t. t and atp.
#include "common.h"

namespace {

struct Mp3BwdArgs {
  const unsigned short* x;    // raw input of the pooling (fp16)
  const unsigned short* dy;   // gradient of the pooled tensor (bf16), N x Ho x Wo x C
  unsigned short* da;         // gradient w.r.t. act(bn(x)) (bf16), N x H x W x C: written
  long long x_sn, x_sh, x_sw, y_sn, y_sh, y_sw, a_sn, a_sh, a_sw;
  int H, W, Ho, Wo, C, groups;
  const float *mean, *var, *gamma, *beta;
  float eps;
  int relu;
  long long total;   // N * H * W * groups
};

__global__ __launch_bounds__(256) void maxpool3s2_bwd_kernel(Mp3BwdArgs a) {
  const long long u = (long long)blockIdx.x * 256 + threadIdx.x;
  if (u >= a.total) return;
  const int g = (int)(u % a.groups);
  long long r = u / a.groups;
  const int xx = (int)(r % a.W);
  r /= a.W;
  const int yy = (int)(r % a.H);
  const long long n = r / a.H;
  float sc[8], sh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = g * 8 + e;
    sc[e] = 1.f, sh[e] = 0.f;
    if (a.mean != nullptr && c < a.C) {
      sc[e] = (a.gamma ? a.gamma[c] : 1.f) / sqrtf(a.var[c] + a.eps);
      sh[e] = (a.beta ? a.beta[c] : 0.f) - a.mean[c] * sc[e];
    }
  }
  auto value = [&](int py, int px) {
    f32x8 t = fd_cvt8<FmtA>(*reinterpret_cast<const u32x4*>(a.x + n * a.x_sn + (long long)py * a.x_sh + (long long)px * a.x_sw + g * 8));
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      t[e] = fmaf(t[e], sc[e], sh[e]);
      if (a.relu) t[e] = fmaxf(t[e], 0.f);
    }
    return t;
  };
  f32x8 acc;
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  // windows (ho, wo) with 2 ho - 1 <= yy <= 2 ho + 1: ho = yy / 2, and (yy + 1) / 2 when yy is odd
  for (int iy = 0; iy < 2; ++iy) {
    const int ho = iy == 0 ? yy / 2 : (yy + 1) / 2;
    if ((iy == 1 && (yy & 1) == 0) || ho >= a.Ho) continue;
    for (int ix = 0; ix < 2; ++ix) {
      const int wo = ix == 0 ? xx / 2 : (xx + 1) / 2;
      if ((ix == 1 && (xx & 1) == 0) || wo >= a.Wo) continue;
      // first maximum of the window in scan order, per channel; `mine`: it is this thread's position
      f32x8 best;
      bool mine[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) best[e] = -3.0e38f, mine[e] = false;
      for (int dy = 0; dy < 3; ++dy) {
        const int py = 2 * ho - 1 + dy;
        if (py < 0 || py >= a.H) continue;
        for (int dx = 0; dx < 3; ++dx) {
          const int px = 2 * wo - 1 + dx;
          if (px < 0 || px >= a.W) continue;
          const f32x8 t = value(py, px);
          const bool me = py == yy && px == xx;
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (t[e] > best[e]) best[e] = t[e], mine[e] = me;
        }
      }
      const f32x8 d = __builtin_convertvector(
          __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(a.dy + n * a.y_sn + (long long)ho * a.y_sh + (long long)wo * a.y_sw + g * 8)), f32x8);
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (mine[e]) acc[e] += d[e];
    }
  }
  *reinterpret_cast<u32x4*>(a.da + n * a.a_sn + (long long)yy * a.a_sh + (long long)xx * a.a_sw + g * 8) =
      __builtin_bit_cast(u32x4, __builtin_convertvector(acc, bf16x8));
}

struct PyrBwdArgs {
  const unsigned short* x;    // fp16, C channels
  const unsigned short* dy;   // bf16, the 4 pyramid channels
  unsigned short* dx;         // bf16, C channels: accumulated into
  long long x_sn, x_sh, x_sw, y_sn, y_sh, y_sw, d_sn, d_sh, d_sw;
  const float* w;   // [4][C]
  const float* b;   // [4]
  float* dw_part;   // [tiles][4][C]
  float* db_part;   // [tiles][4]
  int C, k0, tiles_x, tiles_y;
  float slope;
};

// one workgroup per k0 x k0 tile of one image (the forward kernel's decomposition)
__global__ __launch_bounds__(256) void pyramid_pool4_bwd_kernel(PyrBwdArgs a) {
  extern __shared__ float pyr_lds[];
  const int k0 = a.k0, f = k0 / 8, npx = k0 * k0;
  float* s = pyr_lds;                       // [4][npx] per-pixel dot products, then coef_j[p]
  float* gy = s + 4 * npx;                  // [4][npx] dy per pixel and scale
  float* t = gy + 4 * npx;                  // [4][64] fine-block sums of s
  float* u = t + 256;                       // [4][64] fine-block sums of dy
  float* dl = u + 256;                      // [4][64] per fine block: delta of its window / k_j^2
  float* wl = dl + 256;                     // [4][C]
  const int tid = threadIdx.x;
  const int tile = blockIdx.x % (a.tiles_x * a.tiles_y), n = blockIdx.x / (a.tiles_x * a.tiles_y);
  const int ty = tile / a.tiles_x, tx = tile % a.tiles_x;
  for (int i = tid; i < 4 * a.C; i += 256) wl[i] = a.w[i];
  __syncthreads();
  const unsigned short* xb = a.x + n * a.x_sn + (long long)(ty * k0) * a.x_sh + (long long)(tx * k0) * a.x_sw;
  const unsigned short* yb = a.dy + n * a.y_sn + (long long)(ty * k0) * a.y_sh + (long long)(tx * k0) * a.y_sw;
  unsigned short* db_ = a.dx + n * a.d_sn + (long long)(ty * k0) * a.d_sh + (long long)(tx * k0) * a.d_sw;
  for (int p = tid; p < npx; p += 256) {
    const int py = p / k0, px = p % k0;
    const unsigned short* xp = xb + py * a.x_sh + px * a.x_sw;
    float d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
    for (int c = 0; c < a.C; ++c) {
      const float v = fd_cvt1<FmtA>(xp[c]);
      d0 = fmaf(v, wl[c], d0);
      d1 = fmaf(v, wl[a.C + c], d1);
      d2 = fmaf(v, wl[2 * a.C + c], d2);
      d3 = fmaf(v, wl[3 * a.C + c], d3);
    }
    s[p] = d0, s[npx + p] = d1, s[2 * npx + p] = d2, s[3 * npx + p] = d3;
    const unsigned short* yp = yb + py * a.y_sh + px * a.y_sw;
#pragma unroll
    for (int j = 0; j < 4; ++j) gy[j * npx + p] = fd_cvt1<FmtG>(yp[j]);
  }
  __syncthreads();
  {   // thread (scale j, fine block): sums of its f x f values
    const int j = tid >> 6, bidx = tid & 63, by = bidx >> 3, bx = bidx & 7;
    float acc = 0.f, accg = 0.f;
    for (int r = 0; r < f; ++r)
      for (int q = 0; q < f; ++q) {
        acc += s[j * npx + (by * f + r) * k0 + bx * f + q];
        accg += gy[j * npx + (by * f + r) * k0 + bx * f + q];
      }
    t[j * 64 + bidx] = acc;
    u[j * 64 + bidx] = accg;
  }
  __syncthreads();
  {   // window of g x g fine blocks, g = 8 >> j: z = mean + b, delta = leaky'(z) * sum dy; every fine block learns delta / k_j^2
    const int j = tid >> 6, bidx = tid & 63, by = bidx >> 3, bx = bidx & 7;
    const int g = 8 >> j, wy = by / g * g, wx = bx / g * g;
    float acc = 0.f, accg = 0.f;
    for (int r = 0; r < g; ++r)
      for (int q = 0; q < g; ++q) {
        acc += t[j * 64 + (wy + r) * 8 + wx + q];
        accg += u[j * 64 + (wy + r) * 8 + wx + q];
      }
    const float area = (float)(g * f * g * f);
    const float z = acc / area + a.b[j];
    const float delta = (z > 0.f ? 1.f : a.slope) * accg;
    dl[j * 64 + bidx] = delta / area;
    // db: one term per WINDOW (its first fine block speaks for it)
    s[j * npx + bidx] = (by == wy && bx == wx) ? delta : 0.f;   // s is free from here on (its sums are in t)
  }
  __syncthreads();
  if (tid < 4) {
    float acc = 0.f;
    for (int q = 0; q < 64; ++q) acc += s[tid * npx + q];
    a.db_part[(long long)blockIdx.x * 4 + tid] = acc;
  }
  __syncthreads();
  for (int p = tid; p < npx; p += 256) {   // coef_j[p] and dx[p][c] += sum_j coef_j[p] w_j[c]
    const int py = p / k0, px = p % k0, bidx = (py / f) * 8 + px / f;
    const float c0 = dl[bidx], c1 = dl[64 + bidx], c2 = dl[128 + bidx], c3 = dl[192 + bidx];
    s[p] = c0, s[npx + p] = c1, s[2 * npx + p] = c2, s[3 * npx + p] = c3;
    unsigned short* dp = db_ + py * a.d_sh + px * a.d_sw;
    for (int c = 0; c < a.C; ++c) {
      const float v = fd_cvt1<FmtG>(dp[c]) + c0 * wl[c] + c1 * wl[a.C + c] + c2 * wl[2 * a.C + c] + c3 * wl[3 * a.C + c];
      dp[c] = fd_pk1_sr(v, (unsigned)(dp - a.dx) + (unsigned)c);   // constants over k x k windows on top of a bf16 value: stochastic rounding (common.h)
    }
  }
  __syncthreads();
  for (int jc = tid; jc < 4 * a.C; jc += 256) {   // dw_j[c] = sum_p coef_j[p] x[p][c]: fixed order
    const int j = jc / a.C, c = jc % a.C;
    float acc = 0.f;
    for (int p = 0; p < npx; ++p) acc = fmaf(s[j * npx + p], fd_cvt1<FmtA>(xb[(p / k0) * a.x_sh + (p % k0) * a.x_sw + c]), acc);
    a.dw_part[(long long)blockIdx.x * 4 * a.C + jc] = acc;
  }
}

struct BnDropBwdArgs {
  const unsigned short* x;    // raw values (fp16)
  const unsigned short* dy;   // gradient of the finished values (bf16)
  unsigned short* dx;         // gradient of the raw values (bf16): written
  long long x_sn, x_sh, x_sw, y_sn, y_sh, y_sw, d_sn, d_sh, d_sw;
  int N, H, W, C;
  const float *mean, *var, *gamma, *mask;   // mean NULL: no normalisation (dx = mask * dy)
  float eps;
  float *dgamma, *dbeta;      // [C], written (NULL without a norm)
};

// one workgroup per channel: the tensors are at most 8 x 8 pixels
__global__ __launch_bounds__(256) void bn_dropout_bwd_kernel(BnDropBwdArgs a) {
  __shared__ double sh1[256], sh2[256];
  const int c = blockIdx.x, tid = threadIdx.x;
  const long long M = (long long)a.N * a.H * a.W;
  const float rs = a.mean ? 1.f / sqrtf(a.var[c] + a.eps) : 1.f, mu = a.mean ? a.mean[c] : 0.f;
  double s1 = 0.0, s2 = 0.0;
  for (long long i = tid; i < M; i += 256) {
    const long long n = i / ((long long)a.H * a.W), r = i % ((long long)a.H * a.W);
    const int y = (int)(r / a.W), xx = (int)(r % a.W);
    const float g = fd_cvt1<FmtG>(a.dy[n * a.y_sn + y * a.y_sh + xx * a.y_sw + c]) * (a.mask ? a.mask[n * a.C + c] : 1.f);
    const float xh = (fd_cvt1<FmtA>(a.x[n * a.x_sn + y * a.x_sh + xx * a.x_sw + c]) - mu) * rs;
    s1 += (double)g;
    s2 += (double)g * (double)xh;
  }
  sh1[tid] = s1, sh2[tid] = s2;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) sh1[tid] += sh1[tid + s], sh2[tid] += sh2[tid + s];
    __syncthreads();
  }
  const float S1 = (float)sh1[0], S2 = (float)sh2[0];
  if (tid == 0 && a.mean != nullptr) {
    a.dbeta[c] = S1;
    a.dgamma[c] = S2;
  }
  const float gm = a.mean ? (a.gamma ? a.gamma[c] : 1.f) : 1.f, inv_m = 1.f / (float)M;
  for (long long i = tid; i < M; i += 256) {
    const long long n = i / ((long long)a.H * a.W), r = i % ((long long)a.H * a.W);
    const int y = (int)(r / a.W), xx = (int)(r % a.W);
    const float g = fd_cvt1<FmtG>(a.dy[n * a.y_sn + y * a.y_sh + xx * a.y_sw + c]) * (a.mask ? a.mask[n * a.C + c] : 1.f);
    float v = g;
    if (a.mean != nullptr) {
      const float xh = (fd_cvt1<FmtA>(a.x[n * a.x_sn + y * a.x_sh + xx * a.x_sw + c]) - mu) * rs;
      v = gm * rs * (g - S1 * inv_m - xh * S2 * inv_m);
    }
    a.dx[n * a.d_sn + y * a.d_sh + xx * a.d_sw + c] = fd_pk1<FmtG>(v);
  }
}

struct ScatterBwdArgs {
  const float *x, *tran, *mean;       // forward inputs; mean: the leaky window means A [N][3][nwin]
  const float *g_dehaze2, *g_atp;     // NCHW fp32 gradients of the two outputs (either may be NULL)
  const unsigned short* g_cat;        // NHWC bf16 gradient of the refine input (channels 0-2: J) or NULL
  long long c_sn, c_sh, c_sw;
  float* d_tran;                      // NCHW fp32, written
  float* dA_part;                     // [N][3][nwin][H] row sums of dA
  int H, W, nwin;
  float eps;
};

// one workgroup per (image, channel, window, row): dt per pixel, and the row's sum of dA
__global__ __launch_bounds__(256) void scatter_dehaze_bwd_kernel(ScatterBwdArgs a) {
  __shared__ double sh[256];
  const int row = blockIdx.x % a.H;
  long long r = blockIdx.x / a.H;
  const int win = (int)(r % a.nwin);
  r /= a.nwin;
  const int c = (int)(r % 3);
  const long long n = r / 3, plane = (long long)a.H * a.W;
  const float A = a.mean[(n * 3 + c) * a.nwin + win];
  // columns px with px * nwin / W == win (upsample_nearest of the 1 x nwin map)
  double acc = 0.0;
  for (int px = threadIdx.x; px < a.W; px += 256) {
    if ((int)((long long)px * a.nwin / a.W) != win) continue;
    const long long i = (n * 3 + c) * plane + (long long)row * a.W + px;
    float g = a.g_dehaze2 ? a.g_dehaze2[i] : 0.f;
    if (a.g_cat) g += fd_cvt1<FmtG>(a.g_cat[n * a.c_sn + (long long)row * a.c_sh + (long long)px * a.c_sw + c]);
    const float t = a.tran[i], den = fabsf(t) + a.eps, xv = a.x[i];
    a.d_tran[i] = -g * (xv - A) / (den * den) * (t > 0.f ? 1.f : (t < 0.f ? -1.f : 0.f));
    float dA = g * (1.f - 1.f / den);
    if (a.g_atp) dA += a.g_atp[i];
    acc += (double)dA;
  }
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) a.dA_part[blockIdx.x] = (float)sh[0];
}

struct AtpBwdArgs {
  const float* dA_part;   // [N*3][nwin][H]
  const float* atp;       // forward input [N][3][H][W]: sign of the window mean decides the leaky slope
  const float* mean;      // leaky(window mean): its sign is the mean's sign
  float* d_atp;           // [N][3][H][W], written
  int H, W, nwin;
  float slope;
};

// one workgroup per (image, channel, window): dmean = leaky' * sum of the rows' sums, spread over the H x H window
__global__ __launch_bounds__(256) void atp_window_mean_bwd_kernel(AtpBwdArgs a) {
  __shared__ float dm;
  const int win = blockIdx.x % a.nwin, nc = blockIdx.x / a.nwin;
  if (threadIdx.x == 0) {
    double acc = 0.0;
    for (int r = 0; r < a.H; ++r) acc += (double)a.dA_part[(long long)blockIdx.x * a.H + r];
    const float m = a.mean[blockIdx.x];
    dm = (float)acc * (m > 0.f ? 1.f : a.slope) / ((float)a.H * (float)a.H);
  }
  __syncthreads();
  float* p = a.d_atp + (long long)nc * a.H * a.W;
  // the reference pools windows [win * H, (win + 1) * H) of the W axis (avg_pool2d with kernel H): pixels outside every window
  // (W not a multiple of H) get no gradient
  for (long long i = threadIdx.x; i < (long long)a.H * a.H; i += 256) p[(i / a.H) * a.W + (long long)win * a.H + (i % a.H)] = dm;
}

}  // namespace

extern "C" int fdgan_maxpool3s2_bwd(const FdTensor* x, const FdPrologue* pro, const FdTensor* dy, const FdTensor* da, FdStream stream) {
  FD_REQUIRE(x && dy && da && x->ptr && dy->ptr && da->ptr, "maxpool3s2_bwd: NULL pointer");
  FD_REQUIRE(x->dtype == FD_F16 && dy->dtype == FD_BF16 && da->dtype == FD_BF16, "maxpool3s2_bwd: x fp16, dy / da bf16 NHWC views");
  const int64_t ho = (x->h + 2 - 3) / 2 + 1, wo = (x->w + 2 - 3) / 2 + 1;
  FD_REQUIRE(dy->n == x->n && dy->h == ho && dy->w == wo && dy->c == x->c && da->n == x->n && da->h == x->h && da->w == x->w && da->c == x->c &&
                 x->c % 8 == 0, "maxpool3s2_bwd: shapes");
  FD_REQUIRE((((uintptr_t)x->ptr | (uintptr_t)dy->ptr | (uintptr_t)da->ptr) & 15) == 0, "maxpool3s2_bwd: 16-byte alignment");
  for (int i = 0; i < 3; ++i) FD_REQUIRE(x->stride[i] % 8 == 0 && dy->stride[i] % 8 == 0 && da->stride[i] % 8 == 0, "maxpool3s2_bwd: strides must be multiples of 8");
  FD_REQUIRE(!pro || pro->act == FD_ACT_NONE || pro->act == FD_ACT_RELU, "maxpool3s2_bwd: prologue activation must be NONE or RELU");
  const int groups = (int)(x->c / 8);
  Mp3BwdArgs a{static_cast<const unsigned short*>(x->ptr), static_cast<const unsigned short*>(dy->ptr), static_cast<unsigned short*>(da->ptr),
               x->stride[0], x->stride[1], x->stride[2], dy->stride[0], dy->stride[1], dy->stride[2], da->stride[0], da->stride[1], da->stride[2],
               (int)x->h, (int)x->w, (int)ho, (int)wo, (int)x->c, groups,
               pro ? pro->mean : nullptr, pro ? pro->var : nullptr, pro ? pro->gamma : nullptr, pro ? pro->beta : nullptr, pro ? pro->eps : 0.f,
               pro && pro->act == FD_ACT_RELU ? 1 : 0, x->n * x->h * x->w * groups};
  return fd_launch(&maxpool3s2_bwd_kernel, "maxpool3s2_bwd", dim3((unsigned)((a.total + 255) / 256)), dim3(256), 0, a, static_cast<hipStream_t>(stream));
}

extern "C" int fdgan_pyramid_pool4_bwd(const FdTensor* x, const float* weight, const float* bias, int k0, float slope, const FdTensor* dy,
                                       const FdTensor* dx, float* dw_part, float* db_part, int64_t* tiles_out, FdStream stream) {
  FD_REQUIRE(x && dy && dx && x->ptr && dy->ptr && dx->ptr && weight && bias && dw_part && db_part, "pyramid_pool4_bwd: NULL pointer");
  FD_REQUIRE(x->dtype == FD_F16 && dy->dtype == FD_BF16 && dx->dtype == FD_BF16, "pyramid_pool4_bwd: x fp16, dy / dx bf16 NHWC views");
  FD_REQUIRE(k0 == 16 || k0 == 32, "pyramid_pool4_bwd: largest window %d (16 or 32)", k0);
  FD_REQUIRE(x->n == dy->n && x->h == dy->h && x->w == dy->w && dy->c == 4 && x->c >= 1 && x->c <= 64 && dx->n == x->n && dx->h == x->h &&
                 dx->w == x->w && dx->c == x->c, "pyramid_pool4_bwd: shapes");
  FD_REQUIRE(x->h % k0 == 0 && x->w % k0 == 0, "pyramid_pool4_bwd: %lld x %lld is not a multiple of %d", (long long)x->h, (long long)x->w, k0);
  PyrBwdArgs a{static_cast<const unsigned short*>(x->ptr), static_cast<const unsigned short*>(dy->ptr), static_cast<unsigned short*>(dx->ptr),
               x->stride[0], x->stride[1], x->stride[2], dy->stride[0], dy->stride[1], dy->stride[2], dx->stride[0], dx->stride[1], dx->stride[2],
               weight, bias, dw_part, db_part, (int)x->c, k0, (int)(x->w / k0), (int)(x->h / k0), slope};
  const long long tiles = x->n * a.tiles_x * a.tiles_y;
  if (tiles_out) *tiles_out = tiles;
  const unsigned lds = (8 * k0 * k0 + 768 + 4 * (unsigned)x->c) * 4;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pyramid_pool4_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    if (e != hipSuccess) FD_FAIL(FD_ELAUNCH, "hipFuncSetAttribute(pyramid_pool4_bwd): %s", hipGetErrorString(e));
    attr_done = true;
  }
  return fd_launch(&pyramid_pool4_bwd_kernel, "pyramid_pool4_bwd", dim3((unsigned)tiles), dim3(256), lds, a, static_cast<hipStream_t>(stream));
}

extern "C" int fdgan_bn_dropout_bwd(const FdTensor* x, const float* mean, const float* var, const float* gamma, float eps, const float* mask,
                                    const FdTensor* dy, const FdTensor* dx, float* dgamma, float* dbeta, FdStream stream) {
  FD_REQUIRE(x && dy && dx && x->ptr && dy->ptr && dx->ptr, "bn_dropout_bwd: NULL pointer");
  FD_REQUIRE((mean == nullptr) == (var == nullptr) && (mean == nullptr || (dgamma && dbeta)), "bn_dropout_bwd: mean / var / dgamma / dbeta go together");
  FD_REQUIRE(x->dtype == FD_F16 && dy->dtype == FD_BF16 && dx->dtype == FD_BF16 && x->stride[3] == 1 && dy->stride[3] == 1 && dx->stride[3] == 1,
             "bn_dropout_bwd: x fp16, dy / dx bf16 NHWC views");
  FD_REQUIRE(x->n == dy->n && x->h == dy->h && x->w == dy->w && x->c == dy->c && x->n == dx->n && x->h == dx->h && x->w == dx->w && x->c == dx->c,
             "bn_dropout_bwd: shape mismatch");
  FD_REQUIRE(x->n * x->h * x->w <= (1 << 20), "bn_dropout_bwd: built for the U-Net's innermost levels (one workgroup per channel)");
  BnDropBwdArgs a{static_cast<const unsigned short*>(x->ptr), static_cast<const unsigned short*>(dy->ptr), static_cast<unsigned short*>(dx->ptr),
                  x->stride[0], x->stride[1], x->stride[2], dy->stride[0], dy->stride[1], dy->stride[2], dx->stride[0], dx->stride[1], dx->stride[2],
                  (int)x->n, (int)x->h, (int)x->w, (int)x->c, mean, var, gamma, mask, eps, dgamma, dbeta};
  return fd_launch(&bn_dropout_bwd_kernel, "bn_dropout_bwd", dim3((unsigned)x->c), dim3(256), 0, a, static_cast<hipStream_t>(stream));
}

extern "C" int fdgan_scatter_dehaze_bwd(const float* x, const float* tran, const float* atp, const float* window_mean, int64_t n, int64_t h,
                                        int64_t w, float slope, float eps, const float* g_dehaze2, const float* g_atp, const FdTensor* g_cat,
                                        float* d_tran, float* d_atp, float* scratch, int64_t scratch_floats, FdStream stream) {
  FD_REQUIRE(x && tran && atp && window_mean && d_tran && d_atp && scratch, "scatter_dehaze_bwd: NULL pointer");
  FD_REQUIRE(n > 0 && h > 0 && w >= h, "scatter_dehaze_bwd: W >= H required (dehaze22.py:705)");
  const int nwin = (int)(w / h);
  FD_REQUIRE(scratch_floats >= n * 3 * nwin * h, "scatter_dehaze_bwd: scratch too small (%lld floats needed)", (long long)(n * 3 * nwin * h));
  if (g_cat) FD_REQUIRE(g_cat->ptr && g_cat->dtype == FD_BF16 && g_cat->stride[3] == 1 && g_cat->n == n && g_cat->h == h && g_cat->w == w && g_cat->c >= 3,
                        "scatter_dehaze_bwd: g_cat must be an N x H x W NHWC bf16 view");
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipError_t e = hipMemsetAsync(d_atp, 0, (size_t)(n * 3 * h * w) * sizeof(float), st);
  if (e != hipSuccess) FD_FAIL(FD_ELAUNCH, "hipMemsetAsync: %s", hipGetErrorString(e));
  ScatterBwdArgs a{x, tran, window_mean, g_dehaze2, g_atp, g_cat ? static_cast<const unsigned short*>(g_cat->ptr) : nullptr,
                   g_cat ? g_cat->stride[0] : 0, g_cat ? g_cat->stride[1] : 0, g_cat ? g_cat->stride[2] : 0, d_tran, scratch, (int)h, (int)w, nwin, eps};
  if (int rc = fd_launch(&scatter_dehaze_bwd_kernel, "scatter_dehaze_bwd", dim3((unsigned)(n * 3 * nwin * h)), dim3(256), 0, a, st)) return rc;
  AtpBwdArgs b{scratch, atp, window_mean, d_atp, (int)h, (int)w, nwin, slope};
  return fd_launch(&atp_window_mean_bwd_kernel, "atp_window_mean_bwd", dim3((unsigned)(n * 3 * nwin)), dim3(256), 0, b, st);
}
