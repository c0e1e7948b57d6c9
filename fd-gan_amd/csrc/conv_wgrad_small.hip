// conv_wgrad_small.hip -- weight gradient of the few-channel convolutions at the two ends of the networks:
// FDGAN.conv_refin1 (3 -> 64, 3x3), FDGAN.conv_refin3 (16 -> 3, 3x3), D.layer1 (9 -> 36, 4x4 stride 2)
// (/root/reference/models/dehaze1113.py:744, :749, :196).
//
//   dW[co][ci][ky][kx] = sum over output pixels p of dy[p][co] * a[p * s + (ky, kx) - pad][ci],   db[co] = sum_p dy[p][co]
//
// The per-tap kernel (conv_bwd.hip) treats every tap as its own GEMM with Cin padded to a 64-wide tile: 3 of 64 columns
// used, 370 us for a 1.7 k-element gradient.  Here ALL taps of ALL input channels are the N dimension of ONE GEMM
// (N = Cin * k * k <= 160, + 1 column of ones for the bias), M = Cout <= 64, K = pixels:
//   * 128 output pixels per step; the staging pass writes both operands TRANSPOSED into LDS -- At[co][pixel],
//     Bt[(ci, tap)][pixel], i.e. the im2col patch is formed once, by 2-byte LDS writes, 27..144 per pixel -- so that MFMA
//     fragments are plain 16-byte reads of 8 consecutive pixels (row pitch 272 B: conflict-free);
//   * wave w multiplies the w-th 32-pixel quarter: COT x NT MFMAs per step, accumulators live across all steps of the
//     (persistent) workgroup; each wave leaves one partial result, summed by wgrad_reduce_kernel in a fixed order.
#include "conv_igemm.h"

namespace {

// 2-byte LDS stores that the 16-byte fragment loads (u32x4) must see: without may_alias, type-based alias analysis lets
// the compiler treat them as unrelated to those loads
typedef unsigned short __attribute__((may_alias)) u16a;
typedef unsigned int __attribute__((may_alias)) u32a;

constexpr int WS_PX = 128;                 // pixels per step
constexpr int WS_PITCH = (WS_PX + 8) * 2;  // bytes per LDS row

struct WgSmallArgs {
  const unsigned short* x;
  long long x_sn;
  int x_sh, x_sw;
  const unsigned short* dy;
  long long dy_sn;
  int dy_sh, dy_sw;
  int H, W, Cin, Cout, Ho, Wo, ks, stride, pad;
  long long P;
  int ntiles, Nw, bias;   // Nw = Cin * ks * ks; bias: column Nw of Bt is all ones
  int pro_mode;
  float p_slope, eps;
  const float *p_mean, *p_var, *p_gamma, *p_beta;
  float* part;            // [gridDim.x * 4][Cout * Nw]
  float* bias_part;       // [gridDim.x * 4][Cout]
};

template <int COT, int NT>
__global__ __launch_bounds__(256) void conv_wgrad_small_kernel(WgSmallArgs a) {
  extern __shared__ __attribute__((aligned(16))) char ws_lds[];
  char* At = ws_lds;                               // [COT * 16][WS_PITCH]
  char* Bt = ws_lds + COT * 16 * WS_PITCH;         // [NT * 16][WS_PITCH]
  float* sc_s = reinterpret_cast<float*>(Bt + NT * 16 * WS_PITCH);   // [16] scale, [16] shift
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int KK = a.ks * a.ks, cin8 = (a.Cin + 7) / 8, cout8 = (a.Cout + 7) / 8;
  if (tid < 16) {
    float sc = 1.f, sh = 0.f;
    if (a.pro_mode == 2 && tid < a.Cin) {
      const float g = a.p_gamma ? a.p_gamma[tid] : 1.f, b = a.p_beta ? a.p_beta[tid] : 0.f;
      sc = g / sqrtf(a.p_var[tid] + a.eps);
      sh = b - a.p_mean[tid] * sc;
    }
    sc_s[tid] = sc;
    sc_s[16 + tid] = sh;
  }
  // zero the rows no staging pass writes (tile padding): their products land in outputs that are never stored, but
  // uninitialised LDS may hold NaN patterns that a zero operand does not cancel
  for (int i = tid; i < (COT + NT) * 16 * (WS_PITCH / 4); i += 256) reinterpret_cast<u32a*>(ws_lds)[i] = 0u;
  __syncthreads();

  f32x4 acc[COT][NT];
#pragma unroll
  for (int c = 0; c < COT; ++c)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[c][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int m = lane & 15, kg = lane >> 4;
  const long long HoWo = (long long)a.Ho * a.Wo;
  const int dunits = WS_PX * cout8, xunits = WS_PX * KK * cin8;

  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    // ---- staging: one unit = 8 channels of one pixel (dy) / of one tap of one pixel (x), written transposed
    for (int u = tid; u < dunits + xunits; u += 256) {
      const bool isd = u < dunits;
      const int v = isd ? u : u - dunits;
      const int px = v & (WS_PX - 1), piece = v / WS_PX;
      const long long p = (long long)tile * WS_PX + px;
      const bool pok = p < a.P;
      const long long n = pok ? p / HoWo : 0;
      const int r = pok ? (int)(p - n * HoWo) : 0;
      const int oy = r / a.Wo, ox = r - oy * a.Wo;
      u32x4 raw = {0u, 0u, 0u, 0u};
      if (isd) {
        if (pok) raw = *reinterpret_cast<const u32x4*>(a.dy + n * a.dy_sn + (long long)oy * a.dy_sh + (long long)ox * a.dy_sw + piece * 8);
        // (halves are taken from the 32-bit words: extracting __bf16 ELEMENTS of a bit-cast vector made hipcc 7.2 store word 0's
        // low half for every e)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int co = piece * 8 + e;
          const unsigned short hv = (unsigned short)((raw[e >> 1] >> (16 * (e & 1))) & 0xffffu);
          if (co < COT * 16) *reinterpret_cast<u16a*>(At + co * WS_PITCH + px * 2) = (co < a.Cout) ? hv : (unsigned short)0;
        }
      } else {
        const int tap = piece / cin8, c8 = piece - tap * cin8;
        const int ky = tap / a.ks, kx = tap - ky * a.ks;
        const int iy = oy * a.stride + ky - a.pad, ix = ox * a.stride + kx - a.pad;
        const bool ok = pok && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        if (ok) raw = *reinterpret_cast<const u32x4*>(a.x + n * a.x_sn + (long long)iy * a.x_sh + (long long)ix * a.x_sw + c8 * 8);
        f32x8 f = fd_cvt8<FmtA>(raw);      // the forward input is fp16; the staged operand is bf16 like dy
        if (a.pro_mode != 0) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float t = fmaf(f[e], sc_s[(c8 * 8 + e) & 15], sc_s[16 + ((c8 * 8 + e) & 15)]);
            f[e] = ok ? fmaxf(t, a.p_slope * t) : 0.f;      // zero padding applies to the ACTIVATED input
          }
        }
        const u32x4 hw = fd_pack8<FmtG>(f);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int ci = c8 * 8 + e;
          if (ci < a.Cin)
            *reinterpret_cast<u16a*>(Bt + (ci * KK + tap) * WS_PITCH + px * 2) = (unsigned short)((hw[e >> 1] >> (16 * (e & 1))) & 0xffffu);
        }
      }
    }
    if (a.bias && tid < WS_PX)   // the column of ones (bf16 1.0): db = sum_p dy
      *reinterpret_cast<u16a*>(Bt + a.Nw * WS_PITCH + tid * 2) =
          ((long long)tile * WS_PX + tid < a.P) ? (unsigned short)0x3F80 : (unsigned short)0;
    __syncthreads();
    // ---- this wave's 32-pixel quarter
    bf16x8 af[COT];
#pragma unroll
    for (int c = 0; c < COT; ++c) af[c] = __builtin_bit_cast(bf16x8, lds_read16(At + (c * 16 + m) * WS_PITCH + wave * 64 + kg * 16));
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const bf16x8 bfr = __builtin_bit_cast(bf16x8, lds_read16(Bt + (n * 16 + m) * WS_PITCH + wave * 64 + kg * 16));
#pragma unroll
      for (int c = 0; c < COT; ++c) acc[c][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[c], bfr, acc[c][n], 0, 0, 0);
    }
    __syncthreads();
  }
  // ---- one partial per wave: D lane (l & 15) = column n, rows (l >> 4) * 4 + r = co
  const long long split = (long long)blockIdx.x * 4 + wave;
  const long long numel = (long long)a.Cout * a.Nw;
#pragma unroll
  for (int c = 0; c < COT; ++c)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = c * 16 + kg * 4 + r, col = n * 16 + m;
        if (co >= a.Cout) continue;
        if (col < a.Nw) a.part[split * numel + (long long)co * a.Nw + col] = acc[c][n][r];
        else if (a.bias && col == a.Nw) a.bias_part[split * a.Cout + co] = acc[c][n][r];
      }
}

template <int COT, int NT>
int ws_launch(WgSmallArgs& a, unsigned grid, const char* name, hipStream_t stream) {
  const unsigned lds = (COT + NT) * 16 * WS_PITCH + 32 * 4;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_small_kernel<COT, NT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) FD_FAIL(FD_ELAUNCH, "hipFuncSetAttribute(%s): %s", name, hipGetErrorString(e));
    attr_done = true;
  }
  return fd_launch(&conv_wgrad_small_kernel<COT, NT>, name, dim3(grid), dim3(256), lds, a, stream);
}

}  // namespace

// 0: launched (partials in `workspace`: *nsplit_out arrays of cout * cin * k * k floats, then -- with a bias -- *nsplit_out
// arrays of cout floats at workspace + nsplit * numel); 1: shape not covered / workspace too small (nothing launched); < 0: error
int conv_wgrad_small_launch(const FdTensor* x, const FdTensor* dy, int cout, int ksize, int stride, int pad, int pro_mode, float p_slope,
                            float eps, const float* mean, const float* var, const float* gamma, const float* beta, bool want_bias,
                            float* workspace, long long workspace_floats, long long* nsplit_out, hipStream_t stream) {
  const int cin = (int)x->c, KK = ksize * ksize, Nw = cin * KK, ncol = Nw + (want_bias ? 1 : 0);
  const int cot = (cout + 15) / 16, nt = (ncol + 15) / 16;
  if (cin > 16 || cot > 4 || nt > 10 || workspace == nullptr) return 1;
  WgSmallArgs a{};
  a.x = static_cast<const unsigned short*>(x->ptr), a.x_sn = x->stride[0], a.x_sh = (int)x->stride[1], a.x_sw = (int)x->stride[2];
  a.dy = static_cast<const unsigned short*>(dy->ptr), a.dy_sn = dy->stride[0], a.dy_sh = (int)dy->stride[1], a.dy_sw = (int)dy->stride[2];
  a.H = (int)x->h, a.W = (int)x->w, a.Cin = cin, a.Cout = cout, a.Ho = (int)dy->h, a.Wo = (int)dy->w;
  a.ks = ksize, a.stride = stride, a.pad = pad;
  a.P = (long long)dy->n * dy->h * dy->w;
  a.ntiles = (int)((a.P + WS_PX - 1) / WS_PX);
  a.Nw = Nw, a.bias = want_bias ? 1 : 0;
  a.pro_mode = pro_mode, a.p_slope = p_slope, a.eps = eps, a.p_mean = mean, a.p_var = var, a.p_gamma = gamma, a.p_beta = beta;
  // channel pieces are read as whole 16-byte units: the pixel pitch must cover the padded channel counts
  if (x->stride[2] < (cin + 7) / 8 * 8 || dy->stride[2] < (cout + 7) / 8 * 8) return 1;
  unsigned grid = (unsigned)(a.ntiles < 512 ? a.ntiles : 512);
  const long long numel = (long long)cout * Nw, per = numel + (want_bias ? cout : 0);
  while (grid > 1 && (long long)grid * 4 * per > workspace_floats) grid /= 2;
  if ((long long)grid * 4 * per > workspace_floats) return 1;
  a.part = workspace;
  a.bias_part = want_bias ? workspace + (long long)grid * 4 * numel : nullptr;
  *nsplit_out = (long long)grid * 4;
  // instantiations: the tile counts of the three convs this kernel exists for, rounded up (padding tiles cost MFMAs only)
  if (cot <= 1 && nt <= 10) return ws_launch<1, 10>(a, grid, "conv_wgrad_small_1x10", stream);
  if (cot <= 3 && nt <= 9) return ws_launch<3, 9>(a, grid, "conv_wgrad_small_3x9", stream);
  if (cot <= 4 && nt <= 2) return ws_launch<4, 2>(a, grid, "conv_wgrad_small_4x2", stream);
  if (cot <= 4 && nt <= 10) return ws_launch<4, 10>(a, grid, "conv_wgrad_small_4x10", stream);
  return 1;
}
