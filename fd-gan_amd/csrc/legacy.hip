// legacy.hip -- the two element-wise pieces the DCPDN-era networks of the reference need beside the convolution kernels
// (models/dehaze22.py G :205-362, G2 :364-488, Dense :531-660, dehaze :662-753):
//
//   fdgan_pyramid_pool4   the "1mm" multi-scale head (dehaze22.py:343-356, :634-651): for four window sizes k0, k0/2, k0/4,
//                         k0/8:  avg_pool2d(x, k) -> Conv2d(C, 1, 1) -> LeakyReLU(0.2) -> upsample_nearest(size of x).
//                         The 1x1 conv is linear, so it commutes with the average: per pixel four dot products, block means,
//                         bias, activation, broadcast -- one pass over x, one 8-byte store per pixel, nothing intermediate.
//   fdgan_maxpool3s2_nhwc MaxPool2d(3, 2, 1) of relu(bn(x)): the DenseNet stem's norm0 / relu0 / pool0, with the statistics of what it
//                         stores for the first dense layer's norm1.
//   fdgan_scatter_dehaze  the atmospheric-scattering inversion of `dehaze` (dehaze22.py:699-715): airlight = LeakyReLU of the mean of
//                         G2's output over H x H windows, J = (I - A) / (|t| + 1e-10) + A.
//   fdgan_bn_dropout_nhwc y = mask[n][c] * bn(x): train-mode BatchNorm followed by train-mode Dropout2d (a per-(sample,
//                         channel) mask, dehaze22.py:60-63) on the U-Net's three innermost decoder outputs (at most 8 x 8
//                         pixels): the consumer then sees finished values.
#include "common.h"

namespace {

struct PyrArgs {
  const unsigned short* x;
  long long x_sn, x_sh, x_sw;
  unsigned short* y;
  long long y_sn, y_sh, y_sw;
  const float* w;   // [4][C]
  const float* b;   // [4]
  int C, k0, tiles_x, tiles_y;
  float slope;
};

// one workgroup per k0 x k0 tile of one image; k0 = 8 f, f = the finest window
__global__ __launch_bounds__(256) void pyramid_pool4_kernel(PyrArgs a) {
  extern __shared__ float pyr_lds[];
  float* s = pyr_lds;                       // [4][k0 * k0] per-pixel dot products
  float* t = s + 4 * a.k0 * a.k0;           // [4][64] sums of the f x f blocks
  float* m = t + 256;                       // [4][64] value of the window each fine block lies in, per scale
  float* wl = m + 256;                      // [4][C]
  const int tid = threadIdx.x;
  const int tile = blockIdx.x % (a.tiles_x * a.tiles_y), n = blockIdx.x / (a.tiles_x * a.tiles_y);
  const int ty = tile / a.tiles_x, tx = tile % a.tiles_x;
  const int k0 = a.k0, f = k0 / 8, npx = k0 * k0;
  for (int i = tid; i < 4 * a.C; i += 256) wl[i] = a.w[i];
  __syncthreads();
  const unsigned short* xb = a.x + n * a.x_sn + (long long)(ty * k0) * a.x_sh + (long long)(tx * k0) * a.x_sw;
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
  }
  __syncthreads();
  {   // thread (scale j, fine block b): sum of its f x f values, rows then columns (fixed order)
    const int j = tid >> 6, bidx = tid & 63, by = bidx >> 3, bx = bidx & 7;
    float acc = 0.f;
    for (int r = 0; r < f; ++r)
      for (int q = 0; q < f; ++q) acc += s[j * npx + (by * f + r) * k0 + bx * f + q];
    t[j * 64 + bidx] = acc;
  }
  __syncthreads();
  {   // scale j has windows of g x g fine blocks, g = 8 >> j: every fine block learns its window's mean
    const int j = tid >> 6, bidx = tid & 63, by = bidx >> 3, bx = bidx & 7;
    const int g = 8 >> j, wy = by / g * g, wx = bx / g * g;
    float acc = 0.f;
    for (int r = 0; r < g; ++r)
      for (int q = 0; q < g; ++q) acc += t[j * 64 + (wy + r) * 8 + wx + q];
    const float v = acc / (float)(g * f * g * f) + a.b[j];
    m[j * 64 + bidx] = fmaxf(v, a.slope * v);
  }
  __syncthreads();
  unsigned short* yb = a.y + n * a.y_sn + (long long)(ty * k0) * a.y_sh + (long long)(tx * k0) * a.y_sw;
  for (int p = tid; p < npx; p += 256) {
    const int py = p / k0, px = p % k0, bidx = (py / f) * 8 + px / f;
    *reinterpret_cast<u32x2*>(yb + py * a.y_sh + px * a.y_sw) = fd_pk4<FmtA>((f32x4){m[bidx], m[64 + bidx], m[128 + bidx], m[192 + bidx]});
  }
}

struct BnDropArgs {
  const unsigned short* x;
  unsigned short* y;
  long long x_sn, x_sh, x_sw, y_sn, y_sh, y_sw, h, w;
  const float *mean, *var, *gamma, *beta, *mask;   // per channel (mean NULL: no normalisation); mask [N][C] or NULL
  float eps;
  int C, groups;
  long long total;
};

__global__ void bn_dropout_kernel(BnDropArgs a) {
  const long long u = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= a.total) return;
  long long r = u;
  const int g = (int)(r % a.groups);
  r /= a.groups;
  const long long px = r % a.w;
  r /= a.w;
  const long long py = r % a.h, n = r / a.h;
  const u32x4 v = *reinterpret_cast<const u32x4*>(a.x + n * a.x_sn + py * a.x_sh + px * a.x_sw + g * 8);
  f32x8 f = fd_cvt8<FmtA>(v);
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = g * 8 + e;
    float o = 0.f;
    if (c < a.C) {
      o = f[e];
      if (a.mean != nullptr) {
        const float sc = (a.gamma ? a.gamma[c] : 1.f) / sqrtf(a.var[c] + a.eps);
        o = fmaf(o, sc, (a.beta ? a.beta[c] : 0.f) - a.mean[c] * sc);
      }
      if (a.mask != nullptr) o *= a.mask[n * a.C + c];
    }
    f[e] = o;
  }
  *reinterpret_cast<u32x4*>(a.y + n * a.y_sn + py * a.y_sh + px * a.y_sw + g * 8) = fd_pk8<FmtA>(f);
}

struct Mp3Args {
  const unsigned short* x;
  unsigned short* y;
  long long x_sn, x_sh, x_sw, y_sn, y_sh, y_sw;
  int H, W, Ho, Wo, C, groups;
  const float *mean, *var, *gamma, *beta;   // mean NULL: no normalisation
  float eps;
  int relu;
  float* partial;                           // [blocks][cpad][2] or NULL
  int cpad;
  long long npix;                           // N * Ho * Wo
};

// MaxPool2d(3, 2, 1) of relu(bn(x)) -- torchvision DenseNet's norm0 / relu0 / pool0 (dehaze22.py:540-543).  A thread owns
// (output pixel, 8-channel group); a workgroup = 32 consecutive output pixels x up to 8 groups per pass, and emits one row of
// (sum, sum of squares) of what it stored: the statistics the first dense layer's norm1 needs (fixed summation order).
__global__ __launch_bounds__(256) void maxpool3s2_kernel(Mp3Args a) {
  __shared__ float red[2][32][8][8];
  const int tid = threadIdx.x, pl = tid >> 3, gl = tid & 7;
  const long long p = (long long)blockIdx.x * 32 + pl;
  const bool live = p < a.npix;
  const int wo = live ? (int)(p % a.Wo) : 0, ho = live ? (int)((p / a.Wo) % a.Ho) : 0;
  const long long n = live ? p / ((long long)a.Wo * a.Ho) : 0;
  for (int g0 = 0; g0 < a.groups; g0 += 8) {
    const int g = g0 + gl;
    f32x8 best, s1, s2;
#pragma unroll
    for (int e = 0; e < 8; ++e) best[e] = -3.0e38f, s1[e] = s2[e] = 0.f;
    if (live && g < a.groups) {
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
      for (int dy = 0; dy < 3; ++dy) {
        const int yy = 2 * ho - 1 + dy;
        if (yy < 0 || yy >= a.H) continue;
        for (int dx = 0; dx < 3; ++dx) {
          const int xx = 2 * wo - 1 + dx;
          if (xx < 0 || xx >= a.W) continue;
          const u32x4 v = *reinterpret_cast<const u32x4*>(a.x + n * a.x_sn + (long long)yy * a.x_sh + (long long)xx * a.x_sw + g * 8);
          const f32x8 f = fd_cvt8<FmtA>(v);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float t = fmaf(f[e], sc[e], sh[e]);
            if (a.relu) t = fmaxf(t, 0.f);
            best[e] = fmaxf(best[e], t);
          }
        }
      }
      const u32x4 ob = fd_pk8<FmtA>(best);
      *reinterpret_cast<u32x4*>(a.y + n * a.y_sn + (long long)ho * a.y_sh + (long long)wo * a.y_sw + g * 8) = ob;
      const f32x8 r = fd_cvt8<FmtA>(ob);      // statistics of the STORED (fp16) values
#pragma unroll
      for (int e = 0; e < 8; ++e) s1[e] = r[e], s2[e] = r[e] * r[e];
    }
    if (a.partial != nullptr) {
#pragma unroll
      for (int e = 0; e < 8; ++e) red[0][pl][gl][e] = s1[e], red[1][pl][gl][e] = s2[e];
      __syncthreads();
      if (tid < 128) {     // (which, group, element): sum over the 32 pixels in order
        const int which = tid >> 6, g2 = (tid >> 3) & 7, e = tid & 7;
        float t = 0.f;
        for (int q = 0; q < 32; ++q) t += red[which][q][g2][e];
        const int c = (g0 + g2) * 8 + e;
        if (c < a.cpad) a.partial[((long long)blockIdx.x * a.cpad + c) * 2 + which] = t;
      }
      __syncthreads();
    }
  }
}

struct AtpArgs {
  const float* atp;      // [N][3][H][W]
  float* mean;           // [N][3][W / H]: leaky_relu(mean over the H x H window)
  int H, W, nwin;
  float slope;
};

// one workgroup per (image, channel, H x H window): fixed-order tree reduction
__global__ __launch_bounds__(256) void atp_window_mean_kernel(AtpArgs a) {
  __shared__ double sh[256];
  const int win = blockIdx.x % a.nwin, nc = blockIdx.x / a.nwin;
  const float* p = a.atp + (long long)nc * a.H * a.W + (long long)win * a.H;
  double acc = 0.0;
  for (long long i = threadIdx.x; i < (long long)a.H * a.H; i += 256) acc += (double)p[(i / a.H) * a.W + (i % a.H)];
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float v = (float)(sh[0] / ((double)a.H * a.H));
    a.mean[blockIdx.x] = fmaxf(v, a.slope * v);
  }
}

struct ScatterArgs {
  const float *x, *tran, *mean;
  float *atp_out, *dehaze2;
  unsigned short* cat;
  long long c_sn, c_sh, c_sw;
  int H, W, nwin;
  float eps;
  long long total;   // N * H * W
};

// J = (I - A) / (|t| + eps) + A per pixel and channel (dehaze22.py:699-715), stored as NCHW fp32 (an output of the network) and,
// with the hazy image behind it, as the 6 (+2 zero) channel NHWC fp16 input of refine1
__global__ void scatter_dehaze_kernel(ScatterArgs a) {
  const long long u = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= a.total) return;
  const int px = (int)(u % a.W), py = (int)((u / a.W) % a.H);
  const long long n = u / ((long long)a.W * a.H), plane = (long long)a.H * a.W;
  const int wsel = (int)((long long)px * a.nwin / a.W);      // upsample_nearest of the 1 x nwin map to H x W
  f32x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = 0.f;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const long long i = (n * 3 + c) * plane + (long long)py * a.W + px;
    const float A = a.mean[(n * 3 + c) * a.nwin + wsel];
    const float xv = a.x[i], t = a.tran[i];
    const float d = (xv - A) / (fabsf(t) + a.eps) + A;
    a.atp_out[i] = A;
    a.dehaze2[i] = d;
    o[c] = d;
    o[3 + c] = xv;
  }
  *reinterpret_cast<u32x4*>(a.cat + n * a.c_sn + (long long)py * a.c_sh + (long long)px * a.c_sw) = fd_pk8<FmtA>(o);
}

}  // namespace

extern "C" int fdgan_scatter_dehaze(const float* x, const float* tran, const float* atp, int64_t n, int64_t h, int64_t w, float slope, float eps,
                                    float* window_mean, float* atp_out, float* dehaze2, const FdTensor* cat, FdStream stream) {
  FD_REQUIRE(x && tran && atp && window_mean && atp_out && dehaze2 && cat && cat->ptr, "scatter_dehaze: NULL pointer");
  FD_REQUIRE(n > 0 && h > 0 && w >= h, "scatter_dehaze: the reference pools the airlight over H x H windows (dehaze22.py:705): W >= H required");
  FD_REQUIRE(cat->dtype == FD_F16 && cat->stride[3] == 1 && cat->n == n && cat->h == h && cat->w == w && cat->c >= 8, "scatter_dehaze: cat must be an N x H x W x 8 NHWC fp16 view");
  FD_REQUIRE(((uintptr_t)cat->ptr & 15) == 0 && cat->stride[0] % 8 == 0 && cat->stride[1] % 8 == 0 && cat->stride[2] % 8 == 0, "scatter_dehaze: 16-byte alignment");
  const int nwin = (int)(w / h);
  AtpArgs r{atp, window_mean, (int)h, (int)w, nwin, slope};
  if (int rc = fd_launch(&atp_window_mean_kernel, "atp_window_mean", dim3((unsigned)(n * 3 * nwin)), dim3(256), 0, r, static_cast<hipStream_t>(stream))) return rc;
  ScatterArgs a{x, tran, window_mean, atp_out, dehaze2, static_cast<unsigned short*>(cat->ptr), cat->stride[0], cat->stride[1], cat->stride[2],
                (int)h, (int)w, nwin, eps, n * h * w};
  return fd_launch(&scatter_dehaze_kernel, "scatter_dehaze", dim3((unsigned)((a.total + 255) / 256)), dim3(256), 0, a, static_cast<hipStream_t>(stream));
}

extern "C" int fdgan_maxpool3s2_nhwc(const FdTensor* x, const FdPrologue* pro, const FdTensor* y, float* partial, int64_t capacity_floats,
                                     int64_t* rows_out, FdStream stream) {
  FD_REQUIRE(x && y && x->ptr && y->ptr, "maxpool3s2_nhwc: NULL pointer");
  FD_REQUIRE(x->dtype == FD_F16 && y->dtype == FD_F16 && x->stride[3] == 1 && y->stride[3] == 1, "maxpool3s2_nhwc: NHWC fp16 views required");
  const int64_t ho = (x->h + 2 - 3) / 2 + 1, wo = (x->w + 2 - 3) / 2 + 1;
  FD_REQUIRE(y->n == x->n && y->h == ho && y->w == wo && y->c == x->c && x->c % 8 == 0, "maxpool3s2_nhwc: y must be N x %lld x %lld x C (C %% 8 == 0)",
             (long long)ho, (long long)wo);
  FD_REQUIRE((((uintptr_t)x->ptr | (uintptr_t)y->ptr) & 15) == 0, "maxpool3s2_nhwc: 16-byte alignment");
  for (int i = 0; i < 3; ++i) FD_REQUIRE(x->stride[i] % 8 == 0 && y->stride[i] % 8 == 0, "maxpool3s2_nhwc: strides must be multiples of 8");
  FD_REQUIRE(!pro || !pro->pool2, "maxpool3s2_nhwc: the prologue's 2x2 average pool does not apply here");
  FD_REQUIRE(!pro || pro->act == FD_ACT_NONE || pro->act == FD_ACT_RELU, "maxpool3s2_nhwc: prologue activation must be NONE or RELU");
  const long long npix = x->n * ho * wo, blocks = (npix + 31) / 32;
  const int cpad = (int)x->c;
  if (rows_out) *rows_out = blocks;
  if (partial) FD_REQUIRE(blocks * cpad * 2 <= capacity_floats, "maxpool3s2_nhwc: statistics workspace too small (%lld floats needed)", blocks * cpad * 2);
  Mp3Args a{static_cast<const unsigned short*>(x->ptr), static_cast<unsigned short*>(y->ptr), x->stride[0], x->stride[1], x->stride[2],
            y->stride[0], y->stride[1], y->stride[2], (int)x->h, (int)x->w, (int)ho, (int)wo, (int)x->c, (int)(x->c / 8),
            pro ? pro->mean : nullptr, pro ? pro->var : nullptr, pro ? pro->gamma : nullptr, pro ? pro->beta : nullptr, pro ? pro->eps : 0.f,
            pro && pro->act == FD_ACT_RELU ? 1 : 0, partial, cpad, npix};
  return fd_launch(&maxpool3s2_kernel, "maxpool3s2_nhwc", dim3((unsigned)blocks), dim3(256), 0, a, static_cast<hipStream_t>(stream));
}

extern "C" int fdgan_pyramid_pool4(const FdTensor* x, const float* weight, const float* bias, int k0, float slope, const FdTensor* y,
                                   FdStream stream) {
  FD_REQUIRE(x && y && x->ptr && y->ptr && weight && bias, "pyramid_pool4: NULL pointer");
  FD_REQUIRE(x->dtype == FD_F16 && y->dtype == FD_F16 && x->stride[3] == 1 && y->stride[3] == 1, "pyramid_pool4: NHWC fp16 views required");
  FD_REQUIRE(k0 == 16 || k0 == 32, "pyramid_pool4: largest window %d (16 or 32: windows k0, k0/2, k0/4, k0/8)", k0);
  FD_REQUIRE(x->n == y->n && x->h == y->h && x->w == y->w && y->c == 4 && x->c >= 1 && x->c <= 64, "pyramid_pool4: shapes");
  FD_REQUIRE(x->h % k0 == 0 && x->w % k0 == 0, "pyramid_pool4: %lld x %lld is not a multiple of the largest window %d", (long long)x->h,
             (long long)x->w, k0);
  FD_REQUIRE(((uintptr_t)y->ptr & 7) == 0 && y->stride[2] % 4 == 0 && y->stride[1] % 4 == 0 && y->stride[0] % 4 == 0,
             "pyramid_pool4: the four output channels must be 8-byte aligned");
  PyrArgs a{static_cast<const unsigned short*>(x->ptr), x->stride[0], x->stride[1], x->stride[2],
            static_cast<unsigned short*>(y->ptr), y->stride[0], y->stride[1], y->stride[2], weight, bias,
            (int)x->c, k0, (int)(x->w / k0), (int)(x->h / k0), slope};
  const unsigned lds = (4 * k0 * k0 + 512 + 4 * (unsigned)x->c) * 4;
  return fd_launch(&pyramid_pool4_kernel, "pyramid_pool4", dim3((unsigned)(x->n * a.tiles_x * a.tiles_y)), dim3(256), lds, a,
                   static_cast<hipStream_t>(stream));
}

extern "C" int fdgan_bn_dropout_nhwc(const FdTensor* x, const float* mean, const float* var, const float* gamma, const float* beta, float eps,
                                     const float* mask, const FdTensor* y, FdStream stream) {
  FD_REQUIRE(x && y && x->ptr && y->ptr, "bn_dropout_nhwc: NULL pointer");
  FD_REQUIRE((mean == nullptr) == (var == nullptr), "bn_dropout_nhwc: mean and var go together");
  FD_REQUIRE(x->dtype == FD_F16 && y->dtype == FD_F16 && x->stride[3] == 1 && y->stride[3] == 1, "bn_dropout_nhwc: NHWC fp16 views required");
  FD_REQUIRE(x->n == y->n && x->h == y->h && x->w == y->w && x->c == y->c, "bn_dropout_nhwc: shape mismatch");
  FD_REQUIRE((((uintptr_t)x->ptr | (uintptr_t)y->ptr) & 15) == 0, "bn_dropout_nhwc: 16-byte alignment");
  for (int i = 0; i < 3; ++i) FD_REQUIRE(x->stride[i] % 8 == 0 && y->stride[i] % 8 == 0, "bn_dropout_nhwc: strides must be multiples of 8");
  const int groups = (int)((x->c + 7) / 8);
  BnDropArgs a{static_cast<const unsigned short*>(x->ptr), static_cast<unsigned short*>(y->ptr), x->stride[0], x->stride[1], x->stride[2],
               y->stride[0], y->stride[1], y->stride[2], x->h, x->w, mean, var, gamma, beta, mask, eps, (int)x->c, groups,
               x->n * x->h * x->w * groups};
  return fd_launch(&bn_dropout_kernel, "bn_dropout_nhwc", dim3((unsigned)((a.total + 255) / 256)), dim3(256), 0, a, static_cast<hipStream_t>(stream));
}
