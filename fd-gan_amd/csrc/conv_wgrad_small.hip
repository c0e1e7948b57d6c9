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
//     (persistent) workgroup, whose four waves are summed at the end: one partial result per workgroup, summed by
//     wgrad_reduce_kernel in a fixed order.
#include <stdlib.h>

#include "conv_igemm.h"

namespace {

// 2-byte LDS stores that the 16-byte fragment loads (u32x4) must see: without may_alias, type-based alias analysis lets
// the compiler treat them as unrelated to those loads
typedef unsigned short __attribute__((may_alias)) u16a;
typedef unsigned int __attribute__((may_alias)) u32a;

constexpr int WS_PX = 128;                 // pixels per step
constexpr int WS_PITCH = (WS_PX + 8) * 2;  // bytes per LDS row
constexpr int WS_CH = 10;                  // staging units in flight per thread (one round: 3 -> 64 and 16 -> 3; two: 9 -> 36 4x4)
constexpr int WS_R = 2;                    // rounds: at most 256 * WS_R * WS_CH units per tile (Cout <= 64, Cin <= 16, 4x4)

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
  float* part;            // [gridDim.x][Cout * Nw]
  float* bias_part;       // [gridDim.x][Cout]
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
  int* mtab = reinterpret_cast<int*>(sc_s + 32);   // [2 halves of the workgroup][WS_R * WS_CH] unit descriptors
  if (tid >= 64 && tid < 64 + 2 * WS_R * WS_CH) {
    const int t = tid - 64, j = t % (WS_R * WS_CH);
    const int u = (t / (WS_R * WS_CH)) * WS_PX + 256 * j;
    const int du = WS_PX * cout8, nu = du + WS_PX * KK * cin8, sh8 = cin8 == 2 ? 1 : 0;
    int mt = 0;
    if (u < du) mt = 3 | ((u >> 7) << 12);
    else if (u < nu) {
      const int piece = (u - du) >> 7;
      const int tap = piece >> sh8, c8 = piece - (tap << sh8);
      const int ky = tap / a.ks, kx = tap - ky * a.ks;
      mt = 1 | (ky << 4) | (kx << 8) | (c8 << 12) | (tap << 16);
    }
    mtab[t] = mt;
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

  // Staging units: one unit = 8 channels of one pixel (dy) / of one tap of one pixel (x), written transposed.  Unit u = tid + 256 k
  // belongs to pixel u & 127 = tid & 127 -- the SAME pixel for every unit of a thread, so its (image, row, column) are worked out
  // once per tile -- and to piece u >> 7, which is uniform over a wave and the same in every tile: what a unit is (dy or x, its tap
  // offsets, its channel piece) is decoded ONCE, into scalar registers.  All loads of a round of WS_CH units are issued before
  // the first of them is scattered.  (The first version decoded, loaded and scattered unit by unit inside the tile loop: the full
  // load latency 9 to 19 times per tile and ~1800 instructions per wave and tile, two thirds of them address arithmetic -- 200 us
  // for 50 MB.)
  const int px = tid & (WS_PX - 1);
  const int nunits = dunits + xunits;
  // unit descriptor (mtab, filled below): bit 0 valid, bit 1 dy, bits 4-7 ky, 8-11 kx, 12-15 piece (dy) / c8 (x), 16-23 tap.  Read
  // back from LDS per tile on purpose: decoded into registers ahead of the loop, the 20 descriptors and everything hipcc derives
  // from them took ~1000 spilled scalar registers.
  const int* mrow = mtab + (tid >> 7) * (WS_R * WS_CH);
  const int kkpitch = KK * WS_PITCH;
  char* const at_px = At + px * 2;
  char* const bt_px = Bt + px * 2;
  const unsigned HoWou = (unsigned)HoWo, Wou = (unsigned)a.Wo;
  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const long long p = (long long)tile * WS_PX + px;
    const bool pok = p < a.P;
    const unsigned pu = pok ? (unsigned)p : 0u;
    const unsigned n = pu / HoWou, r = pu - n * HoWou;
    const int oy = (int)(r / Wou), ox = (int)(r - (unsigned)oy * Wou);
    const unsigned short* dyp = a.dy + n * a.dy_sn + (long long)oy * a.dy_sh + (long long)ox * a.dy_sw;
    const unsigned short* ximg = a.x + n * a.x_sn;
    const int iy0 = oy * a.stride - a.pad, ix0 = ox * a.stride - a.pad;
#pragma unroll
    for (int rd = 0; rd < WS_R; ++rd) {
      if (rd > 0 && 256 * rd * WS_CH >= nunits) break;
      u32x4 raw[WS_CH];
      bool okv[WS_CH];
#pragma unroll
      for (int k = 0; k < WS_CH; ++k) {
        // (round 5) every unit loads UNCONDITIONALLY from a clamped address and is zeroed by a select: with the loads inside
        // `if (pok)` / `if (okv)` blocks hipcc drained each one at the end of its block (tools/loop_wait_audit.py: 22 loads per
        // tile followed by s_waitcnt vmcnt(0)) -- the "round of loads in flight" was one load in flight
        const int mt = __builtin_amdgcn_readfirstlane(mrow[rd * WS_CH + k]);
        const bool valid = (mt & 1) != 0, isdy = (mt & 2) != 0;
        const int iy = iy0 + ((mt >> 4) & 15), ix = ix0 + ((mt >> 8) & 15);
        const bool okx = pok && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        okv[k] = valid && (isdy ? pok : okx);
        const unsigned short* srcx = ximg + (okx ? ((long long)iy * a.x_sh + (long long)(ix * a.x_sw + ((mt >> 12) & 15) * 8)) : 0);
        const unsigned short* srcd = dyp + (valid && isdy ? ((mt >> 12) & 15) * 8 : 0);      // dyp itself is clamped to pixel 0 when !pok
        const u32x4 v = *reinterpret_cast<const u32x4*>(valid && !isdy ? srcx : srcd);
        raw[k] = okv[k] ? v : u32x4{0u, 0u, 0u, 0u};
      }
#pragma unroll
      for (int k = 0; k < WS_CH; ++k) {
        const int mt = __builtin_amdgcn_readfirstlane(mrow[rd * WS_CH + k]);
        if (!(mt & 1)) continue;
        if (mt & 2) {
          const int piece = (mt >> 12) & 15;
          char* const row = at_px + piece * 8 * WS_PITCH;
          // (halves are taken from the 32-bit words: extracting __bf16 ELEMENTS of a bit-cast vector made hipcc 7.2 store word 0's
          // low half for every e)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int co = piece * 8 + e;
            const unsigned short hv = (unsigned short)((raw[k][e >> 1] >> (16 * (e & 1))) & 0xffffu);
            if (co < COT * 16) *reinterpret_cast<u16a*>(row + e * WS_PITCH) = (co < a.Cout) ? hv : (unsigned short)0;
          }
        } else {
          const int c8 = (mt >> 12) & 15, tap = (mt >> 16) & 255;
          f32x8 f = fd_cvt8<FmtA>(raw[k]);      // the forward input is fp16; the staged operand is bf16 like dy
          if (a.pro_mode != 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float t = fmaf(f[e], sc_s[(c8 * 8 + e) & 15], sc_s[16 + ((c8 * 8 + e) & 15)]);
              f[e] = okv[k] ? fmaxf(t, a.p_slope * t) : 0.f;      // zero padding applies to the ACTIVATED input
            }
          }
          const u32x4 hw = fd_pack8<FmtG>(f);
          char* const row = bt_px + tap * WS_PITCH + c8 * 8 * kkpitch;
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (c8 * 8 + e < a.Cin)
              *reinterpret_cast<u16a*>(row + e * kkpitch) = (unsigned short)((hw[e >> 1] >> (16 * (e & 1))) & 0xffffu);
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
  // ---- the four waves' sums into wave 0, in a fixed order ((w0 + w1) + w2) + w3, through the staging area (the tile loop's last
  // barrier is behind every wave): ONE partial per workgroup.  (One per wave made the final reduction -- a chain over 2048 partial
  // arrays for 27 workgroups -- longer than this kernel.)
  {
    float* red = reinterpret_cast<float*>(ws_lds);
    for (int w = 1; w < 4; ++w) {
      if (wave == w) {
#pragma unroll
        for (int c = 0; c < COT; ++c)
#pragma unroll
          for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[((c * NT + n) * 4 + r) * 64 + lane] = acc[c][n][r];
      }
      __syncthreads();
      if (wave == 0) {
#pragma unroll
        for (int c = 0; c < COT; ++c)
#pragma unroll
          for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[c][n][r] += red[((c * NT + n) * 4 + r) * 64 + lane];
      }
      __syncthreads();
    }
  }
  if (wave != 0) return;
  // ---- D lane (l & 15) = column n, rows (l >> 4) * 4 + r = co
  const long long split = blockIdx.x;
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
  const unsigned lds = (COT + NT) * 16 * WS_PITCH + 32 * 4 + 2 * WS_R * WS_CH * 4;
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
  if (a.P >= (1ll << 31) || x->stride[2] * (x->w + 1) >= (1ll << 31)) return 1;      // 32-bit pixel arithmetic in the staging pass
  if (WS_PX * ((cout + 7) / 8 + KK * ((cin + 7) / 8)) > 256 * WS_R * WS_CH || ksize > 15) return 1;
  a.ntiles = (int)((a.P + WS_PX - 1) / WS_PX);
  a.Nw = Nw, a.bias = want_bias ? 1 : 0;
  a.pro_mode = pro_mode, a.p_slope = p_slope, a.eps = eps, a.p_mean = mean, a.p_var = var, a.p_gamma = gamma, a.p_beta = beta;
  // channel pieces are read as whole 16-byte units: the pixel pitch must cover the padded channel counts
  if (x->stride[2] < (cin + 7) / 8 * 8 || dy->stride[2] < (cout + 7) / 8 * 8) return 1;
  // resident workgroups per CU: the staging pass is latency- and issue-bound, so as many as registers and LDS allow (3 -> 64: 4)
  static const char* gcap_env = FD_TUNE_GETENV("FDGAN_DEBUG_WGSMALL_GRID");   // tuning aid
  const unsigned lds_bytes = (unsigned)((cot + nt) * 16 * WS_PITCH);
  // (measured: 3 -> 64 180 / 110 / 87 / 82 / 102 us at 256 / 512 / 768 / 1024 / 1536; 9 -> 36 4x4 107 / 74 / 92 / 92 / 108 -- its 27
  // accumulator tiles leave room for two waves per SIMD)
  unsigned gcap = cot * nt > 10 ? 512u : (lds_bytes <= 40 * 1024 ? 1024u : 768u);
  if (gcap_env) gcap = (unsigned)atoi(gcap_env);
  unsigned grid = (unsigned)(a.ntiles < (int)gcap ? a.ntiles : (int)gcap);
  const long long numel = (long long)cout * Nw, per = numel + (want_bias ? cout : 0);
  while (grid > 1 && (long long)grid * per > workspace_floats) grid /= 2;
  if ((long long)grid * per > workspace_floats) return 1;
  a.part = workspace;
  a.bias_part = want_bias ? workspace + (long long)grid * numel : nullptr;
  *nsplit_out = (long long)grid;
  // instantiations: the tile counts of the three convs this kernel exists for, rounded up (padding tiles cost MFMAs only)
  if (cot <= 1 && nt <= 10) return ws_launch<1, 10>(a, grid, "conv_wgrad_small_1x10", stream);
  if (cot <= 3 && nt <= 9) return ws_launch<3, 9>(a, grid, "conv_wgrad_small_3x9", stream);
  if (cot <= 4 && nt <= 2) return ws_launch<4, 2>(a, grid, "conv_wgrad_small_4x2", stream);
  if (cot <= 3 && nt <= 10) return ws_launch<3, 10>(a, grid, "conv_wgrad_small_3x10", stream);
  if (cot <= 4 && nt <= 10) return ws_launch<4, 10>(a, grid, "conv_wgrad_small_4x10", stream);
  return 1;
}
