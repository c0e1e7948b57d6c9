// ssim.hip -- the differentiable SSIM loss term of the training step
// (/root/reference/models/pytorch_ssim/__init__.py:8-73: 11x11 Gaussian window sigma 1.5, depthwise, zero
// padding 5, C1 = 0.01^2, C2 = 0.03^2, mean over everything).  fp32 NCHW planes in and out: both operands are
// the networks' fp32 outputs / targets, and the op is HBM-bound (7 planes read or written per pixel).
//
//   forward   S = ((2 m1 m2 + C1)(2 s12 + C2)) / ((m1^2 + m2^2 + C1)(s1 + s2 + C2)), m = G*x, s1 = G*x^2 - m1^2, ...
//             one 32x32 tile per workgroup: x, y halo tiles in LDS, the five filtered maps by a separable
//             11 + 11 tap pass; writes per-workgroup partial sums of S and the three partial derivatives
//             dS/dm1, dS/dA (A = G*x^2), dS/dB (B = G*xy) the backward needs.
//   backward  dx = w * (G*(dS/dm1) + 2 x G*(dS/dA) + y G*(dS/dB)), w = dL/dS_mean / count  (G is self-adjoint:
//             symmetric window, zero padding).
#include <math.h>

#include "common.h"

namespace {

constexpr int SS_T = 32, SS_R = 5, SS_E = SS_T + 2 * SS_R;   // 32x32 outputs, 42x42 inputs

struct SsimArgs {
  const float *x, *y;          // fwd: the two images; bwd: x, y again
  float *da, *db, *dc;         // fwd: out; bwd: in (as const)
  float* partial;              // fwd: per-workgroup sum of S
  float* dx;                   // bwd: out
  int H, W, tiles_x, tiles_y;
  float g[11];
  float weight;                // bwd: dL/dmean / count
};

__device__ __forceinline__ void ss_load_tile(float (*t)[SS_E + 1], const float* plane, int H, int W, int y0, int x0, int tid) {
  for (int i = tid; i < SS_E * SS_E; i += 256) {
    const int r = i / SS_E, c = i - r * SS_E;
    const int yy = y0 - SS_R + r, xx = x0 - SS_R + c;
    t[r][c] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? plane[(long long)yy * W + xx] : 0.f;   // zero padding
  }
}

__global__ __launch_bounds__(256) void ssim_fwd_kernel(SsimArgs a) {
  __shared__ float tx[SS_E][SS_E + 1], ty[SS_E][SS_E + 1];
  __shared__ float hz[5][SS_E][SS_T + 1];
  __shared__ float red[256];
  const int tid = threadIdx.x, plane = blockIdx.y;
  const int bx = blockIdx.x % a.tiles_x, by = blockIdx.x / a.tiles_x;
  const int x0 = bx * SS_T, y0 = by * SS_T;
  const long long pb = (long long)plane * a.H * a.W;
  ss_load_tile(tx, a.x + pb, a.H, a.W, y0, x0, tid);
  ss_load_tile(ty, a.y + pb, a.H, a.W, y0, x0, tid);
  __syncthreads();
  for (int i = tid; i < SS_E * SS_T; i += 256) {   // horizontal pass of x, y, x^2, y^2, xy
    const int r = i / SS_T, c = i - r * SS_T;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f;
#pragma unroll
    for (int t = 0; t < 11; ++t) {
      const float xv = tx[r][c + t], yv = ty[r][c + t], gw = a.g[t];
      s0 = fmaf(gw, xv, s0);
      s1 = fmaf(gw, yv, s1);
      s2 = fmaf(gw, xv * xv, s2);
      s3 = fmaf(gw, yv * yv, s3);
      s4 = fmaf(gw, xv * yv, s4);
    }
    hz[0][r][c] = s0, hz[1][r][c] = s1, hz[2][r][c] = s2, hz[3][r][c] = s3, hz[4][r][c] = s4;
  }
  __syncthreads();
  float acc = 0.f;
  for (int i = tid; i < SS_T * SS_T; i += 256) {
    const int r = i / SS_T, c = i - r * SS_T;
    const int oy = y0 + r, ox = x0 + c;
    if (oy >= a.H || ox >= a.W) continue;
    float m1 = 0.f, m2 = 0.f, A = 0.f, Bq = 0.f, Cq = 0.f;
#pragma unroll
    for (int t = 0; t < 11; ++t) {
      const float gw = a.g[t];
      m1 = fmaf(gw, hz[0][r + t][c], m1);
      m2 = fmaf(gw, hz[1][r + t][c], m2);
      A = fmaf(gw, hz[2][r + t][c], A);
      Bq = fmaf(gw, hz[3][r + t][c], Bq);
      Cq = fmaf(gw, hz[4][r + t][c], Cq);
    }
    const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
    const float s1 = A - m1 * m1, s2 = Bq - m2 * m2, s12 = Cq - m1 * m2;
    const float N1 = 2.f * m1 * m2 + C1, N2 = 2.f * s12 + C2, D1 = m1 * m1 + m2 * m2 + C1, D2 = s1 + s2 + C2;
    const float inv = 1.f / (D1 * D2), S = N1 * N2 * inv;
    acc += S;
    const long long o = pb + (long long)oy * a.W + ox;
    a.da[o] = 2.f * m2 * (N2 - N1) * inv - S * (2.f * m1 / D1 - 2.f * m1 / D2);   // dS/dm1
    a.db[o] = -S / D2;                                                              // dS/d(G*x^2)
    a.dc[o] = 2.f * N1 * inv;                                                       // dS/d(G*xy)
  }
  red[tid] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) red[tid] += red[tid + s];
    __syncthreads();
  }
  if (tid == 0) a.partial[(long long)blockIdx.y * gridDim.x + blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void ssim_bwd_kernel(SsimArgs a) {
  __shared__ float ta[SS_E][SS_E + 1], tb[SS_E][SS_E + 1], tc[SS_E][SS_E + 1];
  __shared__ float hz[3][SS_E][SS_T + 1];
  const int tid = threadIdx.x, plane = blockIdx.y;
  const int bx = blockIdx.x % a.tiles_x, by = blockIdx.x / a.tiles_x;
  const int x0 = bx * SS_T, y0 = by * SS_T;
  const long long pb = (long long)plane * a.H * a.W;
  ss_load_tile(ta, a.da + pb, a.H, a.W, y0, x0, tid);
  ss_load_tile(tb, a.db + pb, a.H, a.W, y0, x0, tid);
  ss_load_tile(tc, a.dc + pb, a.H, a.W, y0, x0, tid);
  __syncthreads();
  for (int i = tid; i < SS_E * SS_T; i += 256) {
    const int r = i / SS_T, c = i - r * SS_T;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int t = 0; t < 11; ++t) {
      const float gw = a.g[t];
      s0 = fmaf(gw, ta[r][c + t], s0);
      s1 = fmaf(gw, tb[r][c + t], s1);
      s2 = fmaf(gw, tc[r][c + t], s2);
    }
    hz[0][r][c] = s0, hz[1][r][c] = s1, hz[2][r][c] = s2;
  }
  __syncthreads();
  for (int i = tid; i < SS_T * SS_T; i += 256) {
    const int r = i / SS_T, c = i - r * SS_T;
    const int oy = y0 + r, ox = x0 + c;
    if (oy >= a.H || ox >= a.W) continue;
    float ga = 0.f, gb = 0.f, gc = 0.f;
#pragma unroll
    for (int t = 0; t < 11; ++t) {
      const float gw = a.g[t];
      ga = fmaf(gw, hz[0][r + t][c], ga);
      gb = fmaf(gw, hz[1][r + t][c], gb);
      gc = fmaf(gw, hz[2][r + t][c], gc);
    }
    const long long o = pb + (long long)oy * a.W + ox;
    a.dx[o] = a.weight * (ga + 2.f * a.x[o] * gb + a.y[o] * gc);
  }
}

thread_local int g_ssim_window = 11;   // set by the _w entry points around their call into the 11-tap ones
// pytorch_ssim/__init__.py:8-10 gaussian(window_size, 1.5), centred in 11 taps: a window of w < 11 taps with zero padding w / 2
// IS the zero-extended 11-tap window with zero padding 5
void ss_window(float* g) {
  double v[11], s = 0.0;
  const int r = g_ssim_window / 2;
  for (int i = 0; i < 11; ++i) {
    v[i] = (i - 5 >= -r && i - 5 <= r) ? exp(-((i - 5) * (i - 5)) / (2.0 * 1.5 * 1.5)) : 0.0;
    s += v[i];
  }
  for (int i = 0; i < 11; ++i) g[i] = (float)(v[i] / s);
}

}  // namespace

/* S-map statistics of ssim(x, y): partial[planes * tiles] receives per-tile sums of S (mean = sum / (planes*h*w));
 * da, db, dc (planes*h*w floats each) keep what the backward needs.  planes = n*c. */
extern "C" int fdgan_ssim_fwd(const float* x, const float* y, int64_t planes, int64_t h, int64_t w, float* partial,
                              int64_t partial_floats, float* da, float* db, float* dc, FdStream stream) {
  FD_REQUIRE(x && y && partial && da && db && dc, "ssim_fwd: NULL pointer");
  FD_REQUIRE(planes > 0 && planes < 65536 && h > 0 && w > 0, "ssim_fwd: bad sizes");
  SsimArgs a{};
  a.x = x, a.y = y, a.da = da, a.db = db, a.dc = dc, a.partial = partial;
  a.H = (int)h, a.W = (int)w, a.tiles_x = (int)((w + SS_T - 1) / SS_T), a.tiles_y = (int)((h + SS_T - 1) / SS_T);
  FD_REQUIRE(planes * a.tiles_x * a.tiles_y <= partial_floats, "ssim_fwd: partial buffer too small (%lld floats)",
             (long long)planes * a.tiles_x * a.tiles_y);
  ss_window(a.g);
  return fd_launch(&ssim_fwd_kernel, "ssim_fwd", dim3((unsigned)(a.tiles_x * a.tiles_y), (unsigned)planes), dim3(256), 0, a,
                   static_cast<hipStream_t>(stream));
}

/* dx = weight * dS_mean/dx, weight = upstream gradient / (planes*h*w) */
extern "C" int fdgan_ssim_bwd(const float* x, const float* y, const float* da, const float* db, const float* dc,
                              int64_t planes, int64_t h, int64_t w, float weight, float* dx, FdStream stream) {
  FD_REQUIRE(x && y && da && db && dc && dx, "ssim_bwd: NULL pointer");
  FD_REQUIRE(planes > 0 && planes < 65536 && h > 0 && w > 0, "ssim_bwd: bad sizes");
  SsimArgs a{};
  a.x = x, a.y = y, a.da = const_cast<float*>(da), a.db = const_cast<float*>(db), a.dc = const_cast<float*>(dc), a.dx = dx;
  a.H = (int)h, a.W = (int)w, a.tiles_x = (int)((w + SS_T - 1) / SS_T), a.tiles_y = (int)((h + SS_T - 1) / SS_T);
  a.weight = weight;
  ss_window(a.g);
  return fd_launch(&ssim_bwd_kernel, "ssim_bwd", dim3((unsigned)(a.tiles_x * a.tiles_y), (unsigned)planes), dim3(256), 0, a,
                   static_cast<hipStream_t>(stream));
}

/* The same with SSIM(window_size) for odd window sizes <= 11 (pytorch_ssim/__init__.py:39-73 takes the argument; the reference
 * and fdgan_ssim_fwd / _bwd use 11). */
extern "C" int fdgan_ssim_fwd_w(const float* x, const float* y, int64_t planes, int64_t h, int64_t w, int window_size, float* partial,
                                int64_t partial_floats, float* da, float* db, float* dc, FdStream stream) {
  FD_REQUIRE(window_size >= 1 && window_size <= 11 && (window_size & 1), "ssim_fwd_w: window_size %d (odd, <= 11)", window_size);
  g_ssim_window = window_size;
  const int rc = fdgan_ssim_fwd(x, y, planes, h, w, partial, partial_floats, da, db, dc, stream);
  g_ssim_window = 11;
  return rc;
}

extern "C" int fdgan_ssim_bwd_w(const float* x, const float* y, const float* da, const float* db, const float* dc, int64_t planes,
                                int64_t h, int64_t w, int window_size, float weight, float* dx, FdStream stream) {
  FD_REQUIRE(window_size >= 1 && window_size <= 11 && (window_size & 1), "ssim_bwd_w: window_size %d (odd, <= 11)", window_size);
  g_ssim_window = window_size;
  const int rc = fdgan_ssim_bwd(x, y, da, db, dc, planes, h, w, weight, dx, stream);
  g_ssim_window = 11;
  return rc;
}
