// conv_bwd.hip -- backward pieces of the fused convolution:
//
//   forward (conv_igemm.h):   a = act_p(bn(x));  y = conv(a, W) (+ bias);  stored tensor = y
//
//   * data gradient   da = conv^T(dy, W) is a FORWARD convolution of dy with the flipped, transposed filter
//                     (fdgan_pack_conv_weight(..., flip = 1), pad' = k - 1 - pad).  fdgan_conv2d_bwd_data
//                     (conv_igemm.hip, the MK = 1 instantiations of the forward kernel) runs it with the first pass
//                     of the prologue's backward in its epilogue: dpre = da * act'(bn(x)), BatchNorm's two sums,
//                     and optionally dx += gamma * rstd * dpre -- no kernel for it here.
//   * weight gradient dW[co][ci][tap] = sum_px dy[px][co] * a[px + tap][ci]      fdgan_conv2d_bwd_weight
//                     with `a` recomputed from the raw input and the forward prologue (BatchNorm batch
//                     statistics + activation), exactly as the forward staged it (bf16, zero padding).  This file:
//                     the per-tap kernel (1x1 convs, odd shapes) and the split reductions; stride-1 3x3 / 4x4 convs
//                     go to the transpose-read kernels in conv_wgrad_tr.hip.
//   * prologue        the unfused passes, still used behind pooled prologues and strided convs: fdgan_bn_act_bwd
//                     (dpre = da * act'(bn(x)) in place + sums), fdgan_bn_bwd_finalize*, fdgan_bn_bwd_apply
//                     (dx = A * dpre + B * x + C); and the deferred form of the latter's linear part:
//                     fdgan_bn_bwd_coef / fdgan_affine_accumulate.
//   * fdgan_conv2d_bwd_data_direct: any-stride data gradient into an NCHW fp32 tensor (network inputs with 3 / 9 / 16
//                     channels: a thread per pixel, filter in LDS) or an NHWC bf16 view (strided convs in a plan).
// Reference: autograd of nn.Conv2d / nn.BatchNorm2d / nn.LeakyReLU / nn.Sigmoid as composed in
// /root/reference/models/dehaze1113.py:188-230 (D) and :703-801 (FDGAN).
#include <stdlib.h>

#include "conv_igemm.h"

namespace {

// ------------------------------------------------------------------------------------------
// weight gradient.  Workgroup = 64 cout x 64 cin of ONE tap; 4 waves, each 32 x 32 (2 x 2 MFMA
// tiles); k = output pixels, 128 per step.  Both operands are pixel-major in memory with channels
// contiguous, and both need "8 consecutive pixels of one channel" per lane.  The staging pass
// transposes in registers: a thread loads 4 consecutive pixels x 8 channels of each operand (4 x 16 B),
// regroups them with v_perm_b32 into 8 values of 4 pixels x 1 channel and writes those as ds_write_b64
// into the [channel][128 pixels] LDS image (row pitch 272 B), after which the MFMA fragments are plain
// ds_read_b128.  (First version: one pixel per thread and 16 ds_write_b16 per step -- LDS-write bound,
// 35 ms of a 114 ms generator backward.)
// ------------------------------------------------------------------------------------------
constexpr int WG_KPX = 128;                 // pixels per step
constexpr int WG_ROWB = WG_KPX * 2 + 16;    // bytes per channel row (16-byte aligned; 17 x 16 B: conflict-free b128 reads)
constexpr int WG_TILE_B = 64 * WG_ROWB;     // one operand tile

struct WgradArgs {
  const unsigned short* x;    // raw forward input (NHWC fp16)
  long long x_sn;
  int x_sh, x_sw;
  int Hs, Ws, Cin, Cin8;
  const unsigned short* dy;   // gradient of the conv output (NHWC bf16)
  long long dy_sn;
  int dy_sh, dy_sw;
  int Ho, Wo, Cout, Cout8;
  int ks, stride, pad;
  long long P;                // N * Ho * Wo
  // prologue (no side effects in backward)
  int pro_mode;
  float p_slope, eps;
  const float *p_mean, *p_var, *p_gamma, *p_beta;
  float* dw;                  // [Cout][Cin][ks][ks] fp32 (nsplit == 1) or [nsplit][Cout][Cin][ks][ks] partials
  float* dbias;               // [Cout] or NULL ([nsplit][Cout] partials when nsplit > 1)
  int pool;                   // 2x2 average of the activated input (1x1 convs): x is the full-resolution tensor
  int nsplit;                 // pixel splits; partials summed by wgrad_reduce
  long long split_px;         // pixels per split (multiple of 32)
  int tiles_ci, tiles_co;     // 1-D grid: work item -> (split, tap, cout tile, cin tile), cin tile fastest
  int dbg_skip;               // FDGAN_DEBUG_PHASES (results wrong): 1 no global loads, 2 no LDS stores, 4 no fragment reads / MFMAs
};

// 4 k-entries x 8 channels (one u32x4 per entry) -> 8 x (4 entries of one channel), written to rows ch0 .. ch0+7.
// The 16-byte column of a row is XORed with the row's chunk index ((row >> 3) & 7): adjacent lanes hold adjacent
// chunks (rows 8 apart = 32 banks apart at this pitch), and the swizzle spreads them over the banks.
__device__ __forceinline__ void wg_store_transposed(char* tile, int ch0, int px4, const u32x4 (&v)[4]) {
  const int col = (((px4 >> 1) ^ ((ch0 >> 3) & 7)) << 4) | ((px4 & 1) << 3);
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    // dword d holds channels 2d (low half) and 2d+1 (high half) of each entry
    const unsigned lo01 = __builtin_amdgcn_perm(v[1][d], v[0][d], 0x05040100u), lo23 = __builtin_amdgcn_perm(v[3][d], v[2][d], 0x05040100u);
    const unsigned hi01 = __builtin_amdgcn_perm(v[1][d], v[0][d], 0x07060302u), hi23 = __builtin_amdgcn_perm(v[3][d], v[2][d], 0x07060302u);
    *reinterpret_cast<u32x2*>(tile + (ch0 + 2 * d) * WG_ROWB + col) = u32x2{lo01, lo23};
    *reinterpret_cast<u32x2*>(tile + (ch0 + 2 * d + 1) * WG_ROWB + col) = u32x2{hi01, hi23};
  }
}

// T = 64: workgroup tile 64 cout x 64 cin (wave 32 x 32); T = 128: 128 x 128 (wave 64 x 64, 4x the MFMA work for
// 2x the staging: the 1x1 bottleneck / transition shapes, whose dy would otherwise be re-staged by 16 cin tiles).
// POOL: a.pool != 0 (1x1 conv on the 2x2 average of the activated input: the transitions).
// The loads of a step are UNCONDITIONAL (a unit outside the pixel range / the channels / the image reads the tensor's first bytes
// and is zeroed when it is used) and nothing touches their results until the MFMAs of the previous step are issued: with the edge
// cases as branches -- and, pooled, with the four taps averaged right where they were loaded -- every load was waited for at the
// join behind it and the "software pipeline" below prefetched nothing (5 % MFMA busy on the pooled transitions, round 4).
template <int T, int NW, bool POOL>
__global__ __launch_bounds__(64 * NW) void conv_wgrad_kernel(WgradArgs a) {
  constexpr int NTM = T / 32;             // MFMA tiles per wave along cout (2 waves along cout)
  constexpr int NTN = T / (NW / 2) / 16;  // ... along cin (NW / 2 waves along cin)
  constexpr int UPT = T / (16 * NW);      // staging units (4 pixels x 8 channels) per thread and operand
  constexpr int TILE_B = T * WG_ROWB;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char* At = lds;                         // [T ci][128 px]
  char* Dt = lds + TILE_B;                // [T co][128 px]
  float* sc_s = reinterpret_cast<float*>(lds + 2 * TILE_B);   // [T]
  float* sh_s = sc_s + T;
  float* bsum = sh_s + T;                 // [32][T] (dbias only)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // workgroups are dealt to the 8 XCDs round-robin: renumber so that the tiles of one pixel split -- which read the
  // same dy / x pixels -- run on ONE XCD back to back and share its L2 (otherwise every tile pulls its operands
  // through a different L2: (Cin / T) x the dy traffic, (Cout / T) x the x traffic on the fabric)
  int item = blockIdx.x;
  {
    const int per_xcd = gridDim.x >> 3;
    if (item < per_xcd * 8 && !(a.dbg_skip & 8)) item = (item & 7) * per_xcd + (item >> 3);
  }
  const int tci = item % a.tiles_ci;
  int rest = item / a.tiles_ci;
  const int tco = rest % a.tiles_co;
  rest /= a.tiles_co;
  const int kk_ = a.ks * a.ks, tap = rest % kk_, split = rest / kk_;
  const int ci0 = tci * T, co0 = tco * T;
  const int ky = tap / a.ks, kx = tap - ky * a.ks;
  // per-channel scale / shift of this cin tile (BatchNorm fold, as fd_fold_bn but without side effects)
  if (tid < T) {
    const int c = ci0 + tid;
    float sc = 1.f, sh = 0.f;
    if (a.pro_mode == 2) {
      sc = 0.f;
      if (c < a.Cin) {
        const float g = a.p_gamma ? a.p_gamma[c] : 1.f, b = a.p_beta ? a.p_beta[c] : 0.f;
        sc = g / sqrtf(a.p_var[c] + a.eps);
        sh = b - a.p_mean[c] * sc;
      }
    }
    sc_s[tid] = sc;
    sh_s[tid] = sh;
  }
  __syncthreads();
  // staging map: adjacent lanes take adjacent 8-channel chunks of one pixel (T / 8 lanes = one whole 128- or 256-byte
  // run per pixel; the first version gave a wave 64-byte pieces of 16 pixels 2 KB apart and spent 350 of its 410 us
  // waiting for them).  The k order inside a step is free as long as both operands agree: entry 4 q + j of a step is
  // pixel 32 j + q, so load j of a wave covers consecutive pixels.
  constexpr int CH = T / 8;
  static_assert(64 * NW == 4 * T, "one staging unit per thread and operand");
  const int chunk0 = tid % CH, px4 = tid / CH;
  constexpr int CHUNK_STEP = 0;
  const bool want_bias = a.dbias != nullptr && tap == 0 && tci == 0;
  float bs[UPT][8];
#pragma unroll
  for (int u = 0; u < UPT; ++u)
#pragma unroll
    for (int e = 0; e < 8; ++e) bs[u][e] = 0.f;
  const int wco = (wave & 1) * (T / 2), wci = (wave >> 1) * (NTN * 16);
  const int m = lane & 15, g = lane >> 4;
  f32x4 acc[NTM][NTN];
#pragma unroll
  for (int i = 0; i < NTM; ++i)
#pragma unroll
    for (int j = 0; j < NTN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const long long HW = (long long)a.Ho * a.Wo;
  const u32x4 zero4 = {0u, 0u, 0u, 0u};

  const long long p_begin = (long long)split * a.split_px;
  const long long p_end = p_begin + a.split_px < a.P ? p_begin + a.split_px : a.P;
  // software pipeline: the global loads of step s+1 are in flight while the MFMAs of step s run
  constexpr int XT = POOL ? 4 : 1;      // loads per x unit
  u32x4 dv[UPT][4], xv[UPT][4], xl[UPT][4][XT];
  // (image, row, column) of this thread's first pixel, advanced by 128 per step: the 64-bit divisions of the
  // first version (two per pixel and operand) cost more VALU time than the transposition itself
  int q_n, q_oy, q_ox;
  {
    const long long p = p_begin + px4;
    const long long n = p / HW, r = p - n * HW;
    q_n = (int)n, q_oy = (int)(r / a.Wo), q_ox = (int)(r - (long long)q_oy * a.Wo);
  }
  unsigned xokm = 0, dokm = 0;   // bit (4 u + j): the unit holds data (else it is zero: padding, a channel or pixel past the end)
  auto load_step = [&](long long p0) __attribute__((always_inline)) {
    xokm = dokm = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long long p = p0 + 32 * j + px4;
      const bool pin = p < p_end && !(a.dbg_skip & 1);
#pragma unroll
      for (int u = 0; u < UPT; ++u) {
        const int chunk = chunk0 + CHUNK_STEP * u;
        const bool x_ok = ci0 / 8 + chunk < a.Cin8, dy_ok = co0 / 8 + chunk < a.Cout8;
        const bool dok = pin && dy_ok;
        const long long doff = (long long)q_n * a.dy_sn + (long long)q_oy * a.dy_sh + (long long)q_ox * a.dy_sw + co0 + chunk * 8;
        dv[u][j] = *reinterpret_cast<const u32x4*>(a.dy + (dok ? doff : 0));
        const int iy = q_oy * a.stride + ky - a.pad, ix = q_ox * a.stride + kx - a.pad;
        const bool xok = pin && x_ok && (POOL || (iy >= 0 && iy < a.Hs && ix >= 0 && ix < a.Ws));
        const long long xoff = (long long)q_n * a.x_sn + (long long)(POOL ? 2 * iy : iy) * a.x_sh + (long long)(POOL ? 2 * ix : ix) * a.x_sw + ci0 + chunk * 8;
        const unsigned short* src = a.x + (xok ? xoff : 0);
        xl[u][j][0] = *reinterpret_cast<const u32x4*>(src);
        if constexpr (POOL) {
          const int dxs = xok ? a.x_sw : 0, dys = xok ? a.x_sh : 0;
          xl[u][j][1] = *reinterpret_cast<const u32x4*>(src + dxs);
          xl[u][j][2] = *reinterpret_cast<const u32x4*>(src + dys);
          xl[u][j][3] = *reinterpret_cast<const u32x4*>(src + dys + dxs);
        }
        if (dok) dokm |= 1u << (u * 4 + j);
        if (xok) xokm |= 1u << (u * 4 + j);
      }
      q_ox += 32;   // the position of pixel p + 32 (after j = 3: this thread's first pixel of the next step)
      while (q_ox >= a.Wo) {
        q_ox -= a.Wo;
        if (++q_oy == a.Ho) q_oy = 0, ++q_n;
      }
    }
  };
  // what the loaded units become, when they are used: zeros where they hold nothing, the prologue (and the 2x2 average) on x
  auto finish_step = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < UPT; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int chunk = chunk0 + CHUNK_STEP * u;
        const bool dok = (dokm >> (u * 4 + j)) & 1, xok = (xokm >> (u * 4 + j)) & 1;
        u32x4 d = dv[u][j], x;
        if constexpr (POOL) {
          f32x8 f = fd_affine_act(xl[u][j][0], sc_s + chunk * 8, sh_s + chunk * 8, a.p_slope);
          f += fd_affine_act(xl[u][j][1], sc_s + chunk * 8, sh_s + chunk * 8, a.p_slope);
          f += fd_affine_act(xl[u][j][2], sc_s + chunk * 8, sh_s + chunk * 8, a.p_slope);
          f += fd_affine_act(xl[u][j][3], sc_s + chunk * 8, sh_s + chunk * 8, a.p_slope);
          x = fd_pack8<FmtG>(f * 0.25f);   // fp16 in, the bf16 operand out
        } else {
          x = fd_xform8<FmtA, FmtG>(xl[u][j][0], sc_s + chunk * 8, sh_s + chunk * 8, a.pro_mode != 0 ? a.p_slope : 1.f);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) d[q] = dok ? d[q] : 0u, x[q] = xok ? x[q] : 0u;
        dv[u][j] = d, xv[u][j] = x;
      }
  };
  if (p_begin < p_end) load_step(p_begin);
  for (long long p0 = p_begin; p0 < p_end; p0 += WG_KPX) {
    finish_step();
    if (want_bias)
#pragma unroll
      for (int u = 0; u < UPT; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int e = 0; e < 8; ++e) bs[u][e] += __uint_as_float(((dv[u][j][e >> 1] >> ((e & 1) * 16)) & 0xffffu) << 16);
    __syncthreads();   // previous step's fragments consumed
    if (!(a.dbg_skip & 2))
#pragma unroll
    for (int u = 0; u < UPT; ++u) {
      wg_store_transposed(At, (chunk0 + CHUNK_STEP * u) * 8, px4, xv[u]);
      wg_store_transposed(Dt, (chunk0 + CHUNK_STEP * u) * 8, px4, dv[u]);
    }
    __syncthreads();
    if (p0 + WG_KPX < p_end) load_step(p0 + WG_KPX);
    if (!(a.dbg_skip & 4))
#pragma unroll
    for (int sub = 0; sub < WG_KPX / 32; ++sub) {
      bf16x8 af[NTM], bf[NTN];
#pragma unroll
      for (int i = 0; i < NTM; ++i)
        af[i] = __builtin_bit_cast(bf16x8, lds_read16(Dt + (wco + i * 16 + m) * WG_ROWB + (((sub * 4 + g) ^ (((wco + i * 16 + m) >> 3) & 7)) << 4)));   // A: rows = cout
#pragma unroll
      for (int j = 0; j < NTN; ++j)
        bf[j] = __builtin_bit_cast(bf16x8, lds_read16(At + (wci + j * 16 + m) * WG_ROWB + (((sub * 4 + g) ^ (((wci + j * 16 + m) >> 3) & 7)) << 4)));   // B: cols = cin
#pragma unroll
      for (int i = 0; i < NTM; ++i)
#pragma unroll
        for (int j = 0; j < NTN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
  }
  // D layout: column (lane & 15) = cin, rows (lane >> 4) * 4 + r = cout
  const int kk = a.ks * a.ks;
  float* dwp = a.dw + (long long)split * a.Cout * a.Cin * kk;
#pragma unroll
  for (int i = 0; i < NTM; ++i)
#pragma unroll
    for (int j = 0; j < NTN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = co0 + wco + i * 16 + g * 4 + r, ci = ci0 + wci + j * 16 + m;
        if (co < a.Cout && ci < a.Cin) dwp[((long long)co * a.Cin + ci) * kk + tap] = acc[i][j][r];
      }
  if (want_bias) {
    __syncthreads();
#pragma unroll
    for (int u = 0; u < UPT; ++u)
#pragma unroll
      for (int e = 0; e < 8; ++e) bsum[px4 * T + (chunk0 + CHUNK_STEP * u) * 8 + e] = bs[u][e];
    __syncthreads();
    if (tid < T && co0 + tid < a.Cout) {
      float t = 0.f;
      for (int q = 0; q < 32; ++q) t += bsum[q * T + tid];
      a.dbias[(long long)split * a.Cout + co0 + tid] = t;
    }
  }
}

// ------------------------------------------------------------------------------------------
// weight gradient of the dense-layer growth conv (3x3, stride 1, pad 1, Cout <= 32, Cin % 32 == 0), all nine
// taps in one workgroup.  The per-tap kernel above re-stages x and dy for every tap (9x the VALU, LDS and L2
// traffic; 1 ms per 256x256 layer).  Here the k dimension walks image rows: a workgroup owns (image, 128-pixel
// column block, row range, 32-channel slice of Cin); per output row it stages ONE new input row -- transformed
// once, written as three copies shifted by kx so every tap's B fragment is a 16-byte-aligned ds_read_b128 --
// and one dy row; the two previous input rows stay in LDS (3 row slots).  Wave w owns (cout tile w & 1,
// cin tile w >> 1) for all 9 taps: one A read + 9 B reads per 9 MFMAs.
// ------------------------------------------------------------------------------------------
constexpr int W3_PB = 128;                         // pixels per row step
constexpr int W3_PITCH = W3_PB * 2 + 16;           // 272 B per channel row
constexpr int W3_XS_B = 3 * 3 * 32 * W3_PITCH;     // [row slot][kx][32 ci][pitch]
constexpr int W3_DT_B = 32 * W3_PITCH;

struct Wgrad3Args {
  const unsigned short* x;
  long long x_sn;
  int x_sh, x_sw;
  const unsigned short* dy;
  long long dy_sn;
  int dy_sh, dy_sw;
  int H, W, Cin, Cout, Cout8;
  int xblocks, seg_rows, segs;   // column blocks per image; rows per work item; row segments per (image, block)
  int pro_mode;
  float p_slope, eps;
  const float *p_mean, *p_var, *p_gamma, *p_beta;
  float* part;                   // [nsplit][Cout][Cin][9]
};

__global__ __launch_bounds__(256) void conv_wgrad3x3_kernel(Wgrad3Args a) {
  extern __shared__ __attribute__((aligned(16))) char w3_lds[];
  char* Xs = w3_lds;
  char* Dt = w3_lds + W3_XS_B;
  float* sc_s = reinterpret_cast<float*>(Dt + W3_DT_B);
  float* sh_s = sc_s + 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ci0 = blockIdx.x * 32;
  const int item = blockIdx.y;                       // (image, column block, row segment)
  const int seg = item % a.segs, xb = (item / a.segs) % a.xblocks, n = item / (a.segs * a.xblocks);
  const int y_begin = seg * a.seg_rows, y_end = min(a.H, y_begin + a.seg_rows);
  const int xbase = xb * W3_PB;
  if (tid < 32) {
    const int c = ci0 + tid;
    float sc = 1.f, sh = 0.f;
    if (a.pro_mode == 2) {
      const float g = a.p_gamma ? a.p_gamma[c] : 1.f, b = a.p_beta ? a.p_beta[c] : 0.f;
      sc = g / sqrtf(a.p_var[c] + a.eps);
      sh = b - a.p_mean[c] * sc;
    }
    sc_s[tid] = sc;
    sh_s[tid] = sh;
  }
  __syncthreads();
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  const bool is_x = tid < 128;                       // threads 0-127 stage x, 128-255 stage dy
  const int q = (tid & 127) >> 2, chunk = tid & 3;   // 4-pixel group (0..31), 8-channel chunk (0..3)
  const bool dy_ok = chunk < a.Cout8;
  const unsigned short* ximg = a.x + (long long)n * a.x_sn + ci0 + chunk * 8;
  const unsigned short* dimg = a.dy + (long long)n * a.dy_sn + chunk * 8;

  // one input row -> the three kx-shifted transposed copies of slot (row mod 3)
  auto stage_x_row = [&](int row) __attribute__((always_inline)) {
    char* slot = Xs + ((row + 3) % 3) * (3 * 32 * W3_PITCH);
    u32x4 sv[6];
    const bool rok = row >= 0 && row < a.H;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int px = xbase + q * 4 - 1 + j;
      sv[j] = zero4;
      if (rok && px >= 0 && px < a.W) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(ximg + (long long)row * a.x_sh + (long long)px * a.x_sw);
        sv[j] = fd_xform8<FmtA, FmtG>(v, sc_s + chunk * 8, sh_s + chunk * 8, a.pro_mode != 0 ? a.p_slope : 1.f);   // fp16 x -> bf16 operand; padding stays zero
      }
    }
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const u32x4 grp[4] = {sv[kx], sv[kx + 1], sv[kx + 2], sv[kx + 3]};
      char* tile = slot + kx * (32 * W3_PITCH);
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const unsigned lo01 = __builtin_amdgcn_perm(grp[1][d], grp[0][d], 0x05040100u), lo23 = __builtin_amdgcn_perm(grp[3][d], grp[2][d], 0x05040100u);
        const unsigned hi01 = __builtin_amdgcn_perm(grp[1][d], grp[0][d], 0x07060302u), hi23 = __builtin_amdgcn_perm(grp[3][d], grp[2][d], 0x07060302u);
        *reinterpret_cast<u32x2*>(tile + (chunk * 8 + 2 * d) * W3_PITCH + q * 8) = u32x2{lo01, lo23};
        *reinterpret_cast<u32x2*>(tile + (chunk * 8 + 2 * d + 1) * W3_PITCH + q * 8) = u32x2{hi01, hi23};
      }
    }
  };
  auto stage_dy_row = [&](int row) __attribute__((always_inline)) {
    u32x4 dv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int px = xbase + q * 4 + j;
      dv[j] = (dy_ok && px < a.W) ? *reinterpret_cast<const u32x4*>(dimg + (long long)row * a.dy_sh + (long long)px * a.dy_sw) : zero4;
    }
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const unsigned lo01 = __builtin_amdgcn_perm(dv[1][d], dv[0][d], 0x05040100u), lo23 = __builtin_amdgcn_perm(dv[3][d], dv[2][d], 0x05040100u);
      const unsigned hi01 = __builtin_amdgcn_perm(dv[1][d], dv[0][d], 0x07060302u), hi23 = __builtin_amdgcn_perm(dv[3][d], dv[2][d], 0x07060302u);
      *reinterpret_cast<u32x2*>(Dt + (chunk * 8 + 2 * d) * W3_PITCH + q * 8) = u32x2{lo01, lo23};
      *reinterpret_cast<u32x2*>(Dt + (chunk * 8 + 2 * d + 1) * W3_PITCH + q * 8) = u32x2{hi01, hi23};
    }
  };

  const int wco = (wave & 1) * 16, wci = (wave >> 1) * 16;
  const int m = lane & 15, g = lane >> 4;
  f32x4 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nsub = (min(W3_PB, a.W - xbase) + 31) / 32;

  if (is_x) {
    stage_x_row(y_begin - 1);
    stage_x_row(y_begin);
  }
  for (int y = y_begin; y < y_end; ++y) {
    if (is_x)
      stage_x_row(y + 1);        // slot (y+1) % 3: last read two steps ago
    else
      stage_dy_row(y);
    __syncthreads();
    for (int sub = 0; sub < nsub; ++sub) {
      const bf16x8 af = __builtin_bit_cast(bf16x8, lds_read16(Dt + (wco + m) * W3_PITCH + sub * 64 + g * 16));
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const char* slot = Xs + ((y + ky - 1 + 3) % 3) * (3 * 32 * W3_PITCH);
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const bf16x8 bfr = __builtin_bit_cast(bf16x8, lds_read16(slot + kx * (32 * W3_PITCH) + (wci + m) * W3_PITCH + sub * 64 + g * 16));
          acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfr, acc[ky * 3 + kx], 0, 0, 0);
        }
      }
    }
    __syncthreads();             // fragments consumed: the next step overwrites Dt and the oldest x slot
  }
  // D layout: column (lane & 15) = cin, rows (lane >> 4) * 4 + r = cout
  float* dwp = a.part + (long long)item * a.Cout * a.Cin * 9;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = wco + g * 4 + r, ci = ci0 + wci + m;
      if (co < a.Cout) dwp[((long long)co * a.Cin + ci) * 9 + t] = acc[t][r];
    }
}

// partial weight gradients [nsplit][numel] -> out[numel] (+= when accumulate), summed in split order
struct WredArgs {
  const float* part;
  float* out;
  long long numel;
  int nsplit, accumulate;
};
// 64 outputs x 4 split lanes per workgroup, eight loads in flight per thread, fixed summation order.  (One thread
// per output walking all splits took 80 us for 256 splits of a 128 x 224 filter: a quarter of that layer's backward.)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(WredArgs a) {
  __shared__ float sh[4][64];
  const int col = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const long long i = (long long)blockIdx.x * 64 + col;
  float t = 0.f;
  if (i < a.numel) {
    const float* src = a.part + i;
    int s_ = ty;
    for (; s_ + 28 < a.nsplit; s_ += 32) {
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = src[(long long)(s_ + 4 * k) * a.numel];
#pragma unroll
      for (int k = 0; k < 8; ++k) t += v[k];
    }
    for (; s_ < a.nsplit; s_ += 4) t += src[(long long)s_ * a.numel];
  }
  sh[ty][col] = t;
  __syncthreads();
  if (ty == 0 && i < a.numel) {
    t = (sh[0][col] + sh[1][col]) + (sh[2][col] + sh[3][col]);
    a.out[i] = a.accumulate ? a.out[i] + t : t;
  }
}

// Few outputs, many partial rows (bias gradients: 3 .. 288 sums over 300 .. 1024 rows; the few-channel filters): 64 outputs x 16 split
// lanes per workgroup, eight loads in flight per thread, fixed summation order.  With wgrad_reduce_kernel's 4 lanes a 64-element bias
// over 1024 rows was a chain of 32 dependent rounds on ONE workgroup (40-90 us); `pitch`: floats between partial rows (>= numel).
struct WredWideArgs {
  const float* part;
  float* out;
  long long numel, pitch;
  int nsplit, accumulate;
};
__global__ __launch_bounds__(1024) void wgrad_reduce_wide_kernel(WredWideArgs a) {
  __shared__ float sh[16][64];
  const int col = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const long long i = (long long)blockIdx.x * 64 + col;
  float t = 0.f;
  if (i < a.numel) {
    const float* src = a.part + i;
    int s_ = ty;
    for (; s_ + 112 < a.nsplit; s_ += 128) {
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = src[(long long)(s_ + 16 * k) * a.pitch];
#pragma unroll
      for (int k = 0; k < 8; ++k) t += v[k];
    }
    for (; s_ < a.nsplit; s_ += 16) t += src[(long long)s_ * a.pitch];
  }
  sh[ty][col] = t;
  __syncthreads();
  if (ty == 0 && i < a.numel) {
    t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += sh[q][col];
    a.out[i] = a.accumulate ? a.out[i] + t : t;
  }
}

// Batched form: every [nsplit][numel] partial block of a backward walk in ONE launch (job table in device memory, as
// fdgan_pack_conv_weights does for the filter images).  blockIdx.x walks 64-element column groups of all jobs.
struct WredBatchArgs {
  const FdReduceJob* jobs;
  int njobs;
};
__global__ __launch_bounds__(256) void wgrad_reduce_batch_kernel(WredBatchArgs b) {
  __shared__ float sh[4][64];
  const long long g = blockIdx.x;
  int lo = 0, hi = b.njobs - 1;
  while (lo < hi) {   // last job with first_group <= g
    const int mid = (lo + hi + 1) >> 1;
    if (b.jobs[mid].first_group <= g) lo = mid;
    else hi = mid - 1;
  }
  const FdReduceJob j = b.jobs[lo];
  const int col = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const long long i = (g - j.first_group) * 64 + col;
  float t = 0.f;
  if (i < j.numel) {
    const float* src = j.part + i;
    int s_ = ty;
    for (; s_ + 28 < j.nsplit; s_ += 32) {
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = src[(long long)(s_ + 4 * k) * j.numel];
#pragma unroll
      for (int k = 0; k < 8; ++k) t += v[k];
    }
    for (; s_ < j.nsplit; s_ += 4) t += src[(long long)s_ * j.numel];
  }
  sh[ty][col] = t;
  __syncthreads();
  if (ty == 0 && i < j.numel) {
    t = (sh[0][col] + sh[1][col]) + (sh[2][col] + sh[3][col]);      // the summation order of wgrad_reduce_kernel: bitwise the same result
    j.out[i] = j.accumulate ? j.out[i] + t : t;
  }
}

}  // namespace
int fd_wgrad_reduce_wide(const float* part, float* out, long long numel, long long pitch, int nsplit, int accumulate, hipStream_t stream) {
  WredWideArgs r{part, out, numel, pitch, nsplit, accumulate};
  return fd_launch(&wgrad_reduce_wide_kernel, "wgrad_reduce_wide", dim3((unsigned)((numel + 63) / 64)), dim3(1024), 0, r, stream);
}
int fd_wgrad_reduce(const float* part, float* out, long long numel, int nsplit, int accumulate, hipStream_t stream) {
  if (numel <= 4096 && nsplit >= 128) return fd_wgrad_reduce_wide(part, out, numel, numel, nsplit, accumulate, stream);
  WredArgs r{part, out, numel, nsplit, accumulate};
  return fd_launch(&wgrad_reduce_kernel, "wgrad_reduce", dim3((unsigned)((numel + 63) / 64)), dim3(256), 0, r, stream);
}
static int fd_wgrad_reduce_args(const WredArgs& r, hipStream_t st) {   // (picks the 16-lane form for few outputs x many rows)
  return fd_wgrad_reduce(r.part, r.out, r.numel, r.nsplit, r.accumulate, st);
}
extern "C" int fdgan_wgrad_reduce_batch(const FdReduceJob* jobs_device, int64_t njobs, int64_t total_groups, FdStream stream) {
  FD_REQUIRE(jobs_device && njobs > 0 && njobs < (1 << 20) && total_groups > 0 && total_groups < (1ll << 31), "wgrad_reduce_batch: empty job table");
  WredBatchArgs b{jobs_device, (int)njobs};
  return fd_launch(&wgrad_reduce_batch_kernel, "wgrad_reduce_batch", dim3((unsigned)total_groups), dim3(256), 0, b, static_cast<hipStream_t>(stream));
}
namespace {

// ------------------------------------------------------------------------------------------
// prologue backward, pass 1: dpre = da * act'(scale*x + shift) in place (bf16), and per-workgroup
// partial sums (sum dpre, sum dpre * xhat) per channel for BatchNorm's dbeta / dgamma.
// One thread per (pixel, 8-channel group); a workgroup covers 32 pixels x up to 64 channel groups.
// ------------------------------------------------------------------------------------------
struct BnActBwdArgs {
  unsigned short* da;   // in: gradient w.r.t. the activated tensor; out: dpre
  long long da_sn;
  int da_sh, da_sw;
  const unsigned short* x;
  long long x_sn;
  int x_sh, x_sw;
  int H, W, C, C8;
  long long P;
  int pro_mode;
  float slope, eps;
  const float *mean, *var, *gamma, *beta;
  float* partial;   // [rows][cpad][2] or NULL (no norm)
  int cpad;
  // pool != 0 (the transitions' pooled prologue): `da` is the gradient w.r.t. the 2x2-AVERAGED activation at half
  // resolution; pixel (y, x) takes da[y/2][x/2] / 4 (the un-pool), dpre is NOT written back -- only the sums leave
  int pool;
  // pooled form, optional: G (the gradient of x, full resolution) += gamma * rstd * dpre in the same pass; what is left of
  // BatchNorm's backward is then B * x + C per channel (fdgan_bn_bwd_coef), which the caller defers like everywhere else --
  // the second full-resolution pass (fdgan_bn_bwd_apply) is not needed
  unsigned short* dx;
  long long dx_sn;
  int dx_sh, dx_sw;
};

// POOL / DX: a.pool != 0 / a.dx != NULL as compile-time facts, and no branch around a load in the pixel loop -- a lane past the last
// pixel reads its first pixel again and only its stores and sums are predicated.  (With `if (p < a.P) { loads }` hipcc waited for
// every load at the join behind it: 428 us = 1 TB/s on the Fusion-discriminator's 16 x 127 x 127 x 288 BatchNorm input, round 4.)
template <bool POOL, int DX>      // DX: 0 no dx, 1 dx += gamma * rstd * dpre, 2 dx = gamma * rstd * dpre (the buffer's first writer of a walk)
__global__ __launch_bounds__(256) void bn_act_bwd_kernel(BnActBwdArgs a) {
  __shared__ float red[2][32][64];   // [which][pixel slot][channel within this thread column]: reduced below
  const int tid = threadIdx.x;
  const int grp = tid & 7, slot = tid >> 3;   // 8 channel groups (64 channels) x 32 pixels per pass
  const long long HW = (long long)a.H * a.W;
  const unsigned HWu = (unsigned)HW, Wu = (unsigned)a.W;
  {   // one 64-channel chunk per blockIdx.y: the per-channel setup is paid once per workgroup
    const int c8_0 = blockIdx.y * 8;
    const int c8 = c8_0 + grp;
    const bool cok = c8 < a.C8;
    float sc[8], sh[8], xm[8], xr[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = c8 * 8 + e;
      sc[e] = 1.f, sh[e] = 0.f, xm[e] = 0.f, xr[e] = 1.f;
      if (a.pro_mode == 2 && cok && c < a.C) {
        const float rs = 1.f / sqrtf(a.var[c] + a.eps), gmm = a.gamma ? a.gamma[c] : 1.f, b = a.beta ? a.beta[c] : 0.f;
        sc[e] = gmm * rs;
        sh[e] = b - a.mean[c] * sc[e];
        xm[e] = a.mean[c];
        xr[e] = rs;
      }
    }
    float s1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, s2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const long long stride = (long long)gridDim.x * 32;
    for (long long p0 = (long long)blockIdx.x * 32 + slot; cok && p0 < a.P; p0 += 4 * stride) {   // 4 pixels in flight
      u32x4 dvv[4], xvv[4], gvv[4];
      unsigned short* dp[4];
      unsigned short* gp[4];
      bool ok[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const long long pk = p0 + k * stride;
        ok[k] = pk < a.P;
        const long long p = ok[k] ? pk : p0;
        const unsigned pu = (unsigned)p, n = pu / HWu, r = pu - n * HWu;      // 32-bit: a 64-bit division is ~100 VALU instructions,
        const int y = (int)(r / Wu), xx = (int)(r - (unsigned)y * Wu);         // two of them per pixel were most of this kernel
        dp[k] = a.da + n * a.da_sn + (long long)(POOL ? y >> 1 : y) * a.da_sh + (long long)(POOL ? xx >> 1 : xx) * a.da_sw + c8 * 8;
        dvv[k] = *reinterpret_cast<const u32x4*>(dp[k]);
        xvv[k] = *reinterpret_cast<const u32x4*>(a.x + n * a.x_sn + (long long)y * a.x_sh + (long long)xx * a.x_sw + c8 * 8);
        if constexpr (DX != 0) {
          gp[k] = a.dx + n * a.dx_sn + (long long)y * a.dx_sh + (long long)xx * a.dx_sw + c8 * 8;
          if constexpr (DX == 1) gvv[k] = *reinterpret_cast<const u32x4*>(gp[k]);
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const f32x8 d = __builtin_convertvector(__builtin_bit_cast(bf16x8, dvv[k]), f32x8);
        const f32x8 xf = fd_cvt8<FmtA>(xvv[k]);     // the forward input: fp16
        const float live = ok[k] ? 1.f : 0.f;
        f32x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float pre = fmaf(xf[e], sc[e], sh[e]);
          const float gsl = pre > 0.f ? 1.f : a.slope;   // slope 1: identity, 0: ReLU, 0.2: LeakyReLU
          o[e] = d[e] * (POOL ? 0.25f * gsl : gsl) * live;
          s1[e] += o[e];
          s2[e] += o[e] * (xf[e] - xm[e]) * xr[e];
        }
        if (!POOL && ok[k]) *reinterpret_cast<u32x4*>(dp[k]) = __builtin_bit_cast(u32x4, __builtin_convertvector(o, bf16x8));
        if constexpr (DX != 0) {
          f32x8 g = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          if constexpr (DX == 1) g = __builtin_convertvector(__builtin_bit_cast(bf16x8, gvv[k]), f32x8);
#pragma unroll
          for (int e = 0; e < 8; ++e) g[e] = fmaf(sc[e], o[e], g[e]);
          if (ok[k]) *reinterpret_cast<u32x4*>(gp[k]) = __builtin_bit_cast(u32x4, __builtin_convertvector(g, bf16x8));
        }
      }
    }
    if (a.partial != nullptr) {
      __syncthreads();
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        red[0][slot][grp * 8 + e] = s1[e];
        red[1][slot][grp * 8 + e] = s2[e];
      }
      __syncthreads();
      if (tid < 128) {
        const int which = tid >> 6, cl = tid & 63;
        float t = 0.f;
        for (int q = 0; q < 32; ++q) t += red[which][q][cl];
        const int c = c8_0 * 8 + cl;
        if (c < a.cpad) a.partial[((long long)blockIdx.x * a.cpad + c) * 2 + which] = t;
      }
    }
  }
}

// sums -> (dgamma, dbeta): the (sum, sum-of-products) rows of bn_act_bwd reduced in fp64
struct SumFinArgs {
  const float* partial;
  long long rows, cpad, channels;
  float *dbeta, *dgamma;
  int accumulate;
  float *sink_dbeta, *sink_dgamma;   // optional: also added into the parameters' own gradient buffers
  // raw moments: the second sum is sum(dpre * x), turned into sum(dpre * xhat) = rstd * (S2 - mean * S1) here
  const float *raw_mean, *raw_var;
  float raw_eps;
  // two-level reduction of many partial rows (a few thousand pixel tiles x 4 workgroups would take 60 us): with
  // `level1` set, workgroup (x, y) sums the y-th of gridDim.y row slices and writes ONE row of fp32 sums to level1
  float* level1;
  // fused fdgan_bn_bwd_coef (fdgan_bn_bwd_finalize_coef): bsum += B, csum += C of dx = A dpre + B x + C; dbeta / dgamma may be NULL
  const float* coef_gamma;
  float coef_inv_m;
  float *bsum, *csum;
  int coef_store;   // bsum / csum = instead of +=
};
// (An LDS-free form -- 256 threads, the row lanes summed with wave shuffles -- was measured in round 3 because rocprofv3 shows this
// kernel at 24.6 us on average against 3.9 us alone: its workgroups wait for LDS while the side stream's weight-gradient kernels
// hold all 160 KB of every CU.  Without LDS it starts at once -- and the NEXT kernel of the walk waits instead: the step stayed at
// 28.3 ms and the fused bottleneck backward's bracketed time rose from 113 to 125 us.  The wait is the two streams sharing CUs,
// not this kernel.)
__global__ __launch_bounds__(1024) void sum_finalize_kernel(SumFinArgs a) {
  __shared__ double sh[2][32][33];
  const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const long long c = (long long)blockIdx.x * 32 + cl;
  const long long per = (a.rows + gridDim.y - 1) / gridDim.y;
  const long long r0 = (long long)blockIdx.y * per, r1 = r0 + per < a.rows ? r0 + per : a.rows;
  double t1 = 0.0, t2 = 0.0;
  if (c < a.channels) {
    long long r = r0 + rg;
    for (; r + 96 < r1; r += 128) {   // four rows in flight
      const float2 v0 = *reinterpret_cast<const float2*>(a.partial + (r * a.cpad + c) * 2);
      const float2 v1 = *reinterpret_cast<const float2*>(a.partial + ((r + 32) * a.cpad + c) * 2);
      const float2 v2 = *reinterpret_cast<const float2*>(a.partial + ((r + 64) * a.cpad + c) * 2);
      const float2 v3 = *reinterpret_cast<const float2*>(a.partial + ((r + 96) * a.cpad + c) * 2);
      t1 += (double)v0.x, t2 += (double)v0.y;
      t1 += (double)v1.x, t2 += (double)v1.y;
      t1 += (double)v2.x, t2 += (double)v2.y;
      t1 += (double)v3.x, t2 += (double)v3.y;
    }
    for (; r < r1; r += 32) {
      const float2 v = *reinterpret_cast<const float2*>(a.partial + (r * a.cpad + c) * 2);
      t1 += v.x;
      t2 += v.y;
    }
  }
  sh[0][rg][cl] = t1;
  sh[1][rg][cl] = t2;
  __syncthreads();
  if (rg == 0 && c < a.channels) {
    t1 = t2 = 0.0;
    for (int g = 0; g < 32; ++g) {
      t1 += sh[0][g][cl];
      t2 += sh[1][g][cl];
    }
    if (a.level1 != nullptr) {
      a.level1[((long long)blockIdx.y * a.cpad + c) * 2] = (float)t1;
      a.level1[((long long)blockIdx.y * a.cpad + c) * 2 + 1] = (float)t2;
      return;
    }
    if (a.raw_mean != nullptr) t2 = (t2 - (double)a.raw_mean[c] * t1) / sqrt((double)a.raw_var[c] + (double)a.raw_eps);
    if (a.dbeta != nullptr) {
      if (a.accumulate) {
        a.dbeta[c] += (float)t1;
        a.dgamma[c] += (float)t2;
      } else {
        a.dbeta[c] = (float)t1;
        a.dgamma[c] = (float)t2;
      }
    }
    if (a.sink_dbeta != nullptr) a.sink_dbeta[c] += (float)t1;
    if (a.sink_dgamma != nullptr) a.sink_dgamma[c] += (float)t2;
    if (a.bsum != nullptr) {   // same arithmetic as bn_bwd_coef_kernel, on the fp32-rounded sums it would have read
      const float rs = 1.f / sqrtf(a.raw_var[c] + a.raw_eps), gmm = a.coef_gamma ? a.coef_gamma[c] : 1.f;
      const float A = gmm * rs, B = -gmm * rs * rs * (float)t2 * a.coef_inv_m;
      const float Cc = -A * (float)t1 * a.coef_inv_m - B * a.raw_mean[c];
      if (a.coef_store) {
        a.bsum[c] = B;
        a.csum[c] = Cc;
      } else {
        a.bsum[c] += B;
        a.csum[c] += Cc;
      }
    }
  }
}

// prologue backward, pass 2 (BatchNorm only): dx = scale * (dpre - dbeta/M - xhat * dgamma/M), written
// to (or accumulated into) the gradient buffer of x.
struct BnApplyArgs {
  const unsigned short* dpre;
  long long dp_sn;
  int dp_sh, dp_sw;
  const unsigned short* x;
  long long x_sn;
  int x_sh, x_sw;
  unsigned short* dx;
  long long dx_sn;
  int dx_sh, dx_sw;
  int H, W, C, C8;
  long long P;
  float eps, inv_m;
  const float *mean, *var, *gamma, *dbeta, *dgamma;
  int accumulate;
  // pool != 0: `dpre` is the half-resolution gradient w.r.t. the pooled activation; dpre(y, x) = it[y/2][x/2] / 4 *
  // act'(bn(x)) is formed on the fly (the mask needs beta and the activation slope)
  int pool;
  const float* beta;
  float slope;
};
// dx = A[c] * dpre + B[c] * x + C[c] with A = gamma*rstd, B = -gamma*rstd^2*dgamma/M, C = -A*dbeta/M - B*mean:
// a thread keeps one 8-channel group (24 coefficients in registers) and walks pixels with a grid stride.
template <bool POOL, bool ACCUM>      // (no branch around a load in the pixel loop: see bn_act_bwd_kernel)
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(BnApplyArgs a) {
  const int grp = threadIdx.x & 7, slot = threadIdx.x >> 3;   // 8 channel groups x 32 pixels per workgroup pass
  const long long HW = (long long)a.H * a.W;
  const unsigned HWu = (unsigned)HW, Wu = (unsigned)a.W;
  {   // one 64-channel chunk per blockIdx.y
    const int c8 = blockIdx.y * 8 + grp;
    if (c8 >= a.C8) return;
    float A[8], B[8], Cc[8], Sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = c8 * 8 + e;
      A[e] = B[e] = Cc[e] = Sh[e] = 0.f;
      if (c < a.C) {
        const float rs = 1.f / sqrtf(a.var[c] + a.eps), gmm = a.gamma ? a.gamma[c] : 1.f;
        A[e] = gmm * rs;
        B[e] = -gmm * rs * rs * a.dgamma[c] * a.inv_m;
        Cc[e] = -A[e] * a.dbeta[c] * a.inv_m - B[e] * a.mean[c];
        Sh[e] = (a.beta ? a.beta[c] : 0.f) - a.mean[c] * A[e];     // bn(x) = A x + Sh
      }
    }
    // four pixels per iteration: 8-12 independent 16-byte loads in flight per thread (the single-pixel loop ran
    // at 2.7 TB/s of its 4-tensor traffic)
    const long long stride = (long long)gridDim.x * 32;
    for (long long p0 = (long long)blockIdx.x * 32 + slot; p0 < a.P; p0 += 4 * stride) {
      u32x4 dv[4], xv[4], gv[4];
      unsigned short* op[4];
      bool ok[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const long long pk = p0 + k * stride;
        ok[k] = pk < a.P;
        const long long p = ok[k] ? pk : p0;
        const unsigned pu = (unsigned)p, n = pu / HWu, r = pu - n * HWu;      // 32-bit: a 64-bit division is ~100 VALU instructions,
        const int y = (int)(r / Wu), xx = (int)(r - (unsigned)y * Wu);         // two of them per pixel were most of this kernel
        dv[k] = *reinterpret_cast<const u32x4*>(a.dpre + n * a.dp_sn + (long long)(POOL ? y >> 1 : y) * a.dp_sh +
                                                (long long)(POOL ? xx >> 1 : xx) * a.dp_sw + c8 * 8);
        xv[k] = *reinterpret_cast<const u32x4*>(a.x + n * a.x_sn + (long long)y * a.x_sh + (long long)xx * a.x_sw + c8 * 8);
        op[k] = a.dx + n * a.dx_sn + (long long)y * a.dx_sh + (long long)xx * a.dx_sw + c8 * 8;
        if constexpr (ACCUM) gv[k] = *reinterpret_cast<const u32x4*>(op[k]);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const f32x8 d = __builtin_convertvector(__builtin_bit_cast(bf16x8, dv[k]), f32x8);
        const f32x8 xf = fd_cvt8<FmtA>(xv[k]);      // the forward input: fp16
        f32x8 o;
        if constexpr (ACCUM) o = __builtin_convertvector(__builtin_bit_cast(bf16x8, gv[k]), f32x8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float de = d[e];
          if constexpr (POOL) de *= fmaf(A[e], xf[e], Sh[e]) > 0.f ? 0.25f : 0.25f * a.slope;
          const float v = fmaf(A[e], de, fmaf(B[e], xf[e], Cc[e]));
          o[e] = ACCUM ? o[e] + v : v;
        }
        if (ok[k]) *reinterpret_cast<u32x4*>(op[k]) = __builtin_bit_cast(u32x4, __builtin_convertvector(o, bf16x8));
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// direct data gradient into NCHW fp32 (network inputs): dx[n][ci][iy][ix] = sum over (co, taps) of
// dy[n][oy][ox][co] * W[co][ci][ky][kx] with oy*stride + ky - pad == iy.  One thread per dx element.
// ------------------------------------------------------------------------------------------
struct DgradDirectArgs {
  const unsigned short* dy;
  long long dy_sn;
  int dy_sh, dy_sw;
  int Ho, Wo, Cout;
  const float* w;   // [Cout][Cin][ks][ks] fp32
  float* dx;        // [N][Cin][H][W] fp32, or NULL when dx_nhwc is used
  unsigned short* dx_nhwc;   // NHWC bf16 view (strided convs inside a plan)
  long long dxn_sn;
  int dxn_sh, dxn_sw;
  int N, Cin, H, W, ks, stride, pad;
};
__global__ __launch_bounds__(256) void dgrad_direct_kernel(DgradDirectArgs a) {
  const long long u = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)a.N * a.Cin * a.H * a.W;
  if (u >= total) return;
  const int ix = (int)(u % a.W);
  long long r = u / a.W;
  const int iy = (int)(r % a.H);
  r /= a.H;
  const int ci = (int)(r % a.Cin), n = (int)(r / a.Cin);
  float s = 0.f;
  for (int ky = 0; ky < a.ks; ++ky) {
    const int ty = iy + a.pad - ky;
    if (ty < 0 || ty % a.stride) continue;
    const int oy = ty / a.stride;
    if (oy >= a.Ho) continue;
    for (int kx = 0; kx < a.ks; ++kx) {
      const int tx = ix + a.pad - kx;
      if (tx < 0 || tx % a.stride) continue;
      const int ox = tx / a.stride;
      if (ox >= a.Wo) continue;
      const unsigned short* dp = a.dy + (long long)n * a.dy_sn + (long long)oy * a.dy_sh + (long long)ox * a.dy_sw;
      for (int co = 0; co < a.Cout; ++co)
        s = fmaf(__uint_as_float((unsigned)dp[co] << 16),
                 (float)(_Float16)a.w[(((long long)co * a.Cin + ci) * a.ks + ky) * a.ks + kx], s);   // the forward's fp16 filter
    }
  }
  if (a.dx != nullptr)
    a.dx[u] = s;
  else
    a.dx_nhwc[(long long)n * a.dxn_sn + (long long)iy * a.dxn_sh + (long long)ix * a.dxn_sw + ci] =
        (unsigned short)(__builtin_bit_cast(unsigned short, (__bf16)s));
}

// Same gradient, one thread per input PIXEL with all CIN channels in registers (network inputs have 3 / 9 / 16
// channels): every dy vector is loaded once per tap instead of once per (tap, channel, filter), and the filter --
// rounded to fp16 as the forward used it -- sits in LDS as [tap][ci][co].  (The per-element kernel above took 2.4 ms
// for the discriminator's 9-channel 256x256 input.)
template <int CIN>
__global__ __launch_bounds__(256) void dgrad_direct_px_kernel(DgradDirectArgs a) {
  extern __shared__ __attribute__((aligned(16))) char dd_lds[];
  float* w_s = reinterpret_cast<float*>(dd_lds);   // [ks*ks][CIN][co8 * 8]
  const int co8 = (a.Cout + 7) / 8, cop = co8 * 8, kk = a.ks * a.ks;
  for (int i = threadIdx.x; i < kk * CIN * cop; i += 256) {
    const int co = i % cop, ci = (i / cop) % CIN, tap = i / (cop * CIN);
    w_s[i] = co < a.Cout ? (float)(_Float16)a.w[((long long)co * CIN + ci) * kk + tap] : 0.f;
  }
  __syncthreads();
  const long long u = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)a.N * a.H * a.W;
  if (u >= total) return;
  const int ix = (int)(u % a.W);
  const long long r = u / a.W;
  const int iy = (int)(r % a.H), n = (int)(r / a.H);
  float acc[CIN];
#pragma unroll
  for (int ci = 0; ci < CIN; ++ci) acc[ci] = 0.f;
  for (int ky = 0; ky < a.ks; ++ky) {
    const int ty = iy + a.pad - ky;
    if (ty < 0 || ty % a.stride) continue;
    const int oy = ty / a.stride;
    if (oy >= a.Ho) continue;
    for (int kx = 0; kx < a.ks; ++kx) {
      const int tx = ix + a.pad - kx;
      if (tx < 0 || tx % a.stride) continue;
      const int ox = tx / a.stride;
      if (ox >= a.Wo) continue;
      const unsigned short* dp = a.dy + (long long)n * a.dy_sn + (long long)oy * a.dy_sh + (long long)ox * a.dy_sw;
      const float* wt = w_s + (ky * a.ks + kx) * CIN * cop;
      for (int c = 0; c < co8; ++c) {
        f32x8 d = __builtin_convertvector(__builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(dp + c * 8)), f32x8);
        if (c == co8 - 1)   // channels past Cout of the last vector are whatever the buffer holds
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (c * 8 + e >= a.Cout) d[e] = 0.f;
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) {
          const f32x4 w0 = *reinterpret_cast<const f32x4*>(wt + ci * cop + c * 8), w1 = *reinterpret_cast<const f32x4*>(wt + ci * cop + c * 8 + 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[ci] = fmaf(d[e], w0[e], fmaf(d[e + 4], w1[e], acc[ci]));
        }
      }
    }
  }
  const long long plane = (long long)a.H * a.W;
  float* o = a.dx + (long long)n * CIN * plane + (long long)iy * a.W + ix;
#pragma unroll
  for (int ci = 0; ci < CIN; ++ci) o[ci * plane] = acc[ci];
}

// output-activation backward: g[n][y][x][c] (NHWC bf16) = dout[n][c][y][x] * f'(out) from NCHW fp32 tensors
// (the sigmoid map D returns, the tanh image FDGAN returns); channels c >= C of the view are zeroed.
struct OutActBwdArgs {
  const float *dout, *out;
  unsigned short* g;
  long long g_sn;
  int g_sh, g_sw;
  int N, C, H, W, act, C8;
};
__global__ __launch_bounds__(256) void out_act_bwd_kernel(OutActBwdArgs a) {
  const unsigned u = blockIdx.x * 256u + threadIdx.x;             // N * H * W * C8 < 2^31 (launcher): 32-bit index arithmetic
  if (u >= (unsigned)a.N * a.H * a.W * a.C8) return;
  const int c8 = (int)(u % (unsigned)a.C8);
  unsigned r = u / (unsigned)a.C8;
  const int x = (int)(r % (unsigned)a.W);
  r /= (unsigned)a.W;
  const int y = (int)(r % (unsigned)a.H), n = (int)(r / (unsigned)a.H);
  f32x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = c8 * 8 + e;
    float v = 0.f;
    if (c < a.C) {
      const long long i = (((long long)n * a.C + c) * a.H + y) * a.W + x;
      const float t = a.out[i];
      v = a.dout[i] * (a.act == FD_ACT_SIGMOID ? t * (1.f - t) : (a.act == FD_ACT_TANH ? 1.f - t * t : 1.f));
    }
    o[e] = v;
  }
  *reinterpret_cast<u32x4*>(a.g + (long long)n * a.g_sn + (long long)y * a.g_sh + (long long)x * a.g_sw + c8 * 8) =
      __builtin_bit_cast(u32x4, __builtin_convertvector(o, bf16x8));
}

// elementwise gradient plumbing on NHWC bf16 views (one thread per pixel x 8-channel group of dst):
//   mode 0  dst += src                                   (a tensor with several consumers; torch.cat / copies)
//   mode 1  dst  = 0.25 * src[y/2][x/2]                  (backward of the 2x2 average pool in a prologue)
//   mode 2  dst  = sum of the 2x2 block of src           (backward of the nearest x2 upsample epilogue)
//   mode 3  dst  = src where ref > 0, else 0             (ReLU epilogue, ref = the stored post-activation output)
struct GradEwArgs {
  const unsigned short *src, *ref;
  unsigned short* dst;
  long long s_sn, r_sn, d_sn;
  int s_sh, s_sw, r_sh, r_sw, d_sh, d_sw;
  int N, H, W, C8, mode, C;   // H, W: dst size; C: channels of the view
};
__global__ __launch_bounds__(256) void grad_ew_kernel(GradEwArgs a) {
  const unsigned u = blockIdx.x * 256u + threadIdx.x;             // N * H * W * C8 < 2^31 (launcher): 32-bit index arithmetic
  if (u >= (unsigned)a.N * a.H * a.W * a.C8) return;
  const int c8 = (int)(u % (unsigned)a.C8);
  unsigned r = u / (unsigned)a.C8;
  const int x = (int)(r % (unsigned)a.W);
  r /= (unsigned)a.W;
  const int y = (int)(r % (unsigned)a.H), n = (int)(r / (unsigned)a.H);
  auto ld = [&](const unsigned short* base, long long sn, int sh, int sw, int yy, int xx) {
    return __builtin_convertvector(
        __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(base + n * sn + (long long)yy * sh + (long long)xx * sw + c8 * 8)), f32x8);
  };
  unsigned short* dp = a.dst + n * a.d_sn + (long long)y * a.d_sh + (long long)x * a.d_sw + c8 * 8;
  f32x8 o;
  if (a.mode == 0) {
    o = __builtin_convertvector(__builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(dp)), f32x8) + ld(a.src, a.s_sn, a.s_sh, a.s_sw, y, x);
  } else if (a.mode == 1) {
    o = ld(a.src, a.s_sn, a.s_sh, a.s_sw, y >> 1, x >> 1) * 0.25f;
  } else if (a.mode == 2) {
    o = (ld(a.src, a.s_sn, a.s_sh, a.s_sw, 2 * y, 2 * x) + ld(a.src, a.s_sn, a.s_sh, a.s_sw, 2 * y, 2 * x + 1)) +
        (ld(a.src, a.s_sn, a.s_sh, a.s_sw, 2 * y + 1, 2 * x) + ld(a.src, a.s_sn, a.s_sh, a.s_sw, 2 * y + 1, 2 * x + 1));
  } else {
    const f32x8 s_ = ld(a.src, a.s_sn, a.s_sh, a.s_sw, y, x);
    const f32x8 rf = fd_cvt8<FmtA>(*reinterpret_cast<const u32x4*>(a.ref + n * a.r_sn + (long long)y * a.r_sh + (long long)x * a.r_sw + c8 * 8));   // a stored activation: fp16
    const float neg = a.mode == 4 ? 0.2f : 0.f;       // 3: ReLU epilogue, 4: LeakyReLU(0.2) epilogue (sign(out) == sign(pre-activation))
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (c8 * 8 + e >= a.C || rf[e] > 0.f) ? s_[e] : neg * s_[e];   // channels past C: untouched
  }
  *reinterpret_cast<u32x4*>(dp) = __builtin_bit_cast(u32x4, __builtin_convertvector(o, bf16x8));
}

// dtype: FD_F16 for a forward activation, FD_BF16 for a gradient
int check_view(const FdTensor* t, const char* what, int dtype = FD_BF16) {
  FD_REQUIRE(t && t->ptr, "%s: NULL tensor", what);
  FD_REQUIRE(t->dtype == dtype, "%s: %s view expected (dtype %d)", what, dtype == FD_F16 ? "an NHWC fp16 activation" : "an NHWC bf16 gradient", t->dtype);
  FD_REQUIRE(t->stride[3] == 1 && t->stride[2] % 8 == 0 && t->stride[1] % 8 == 0 && t->stride[0] % 8 == 0 &&
                 ((uintptr_t)t->ptr & 15) == 0,
             "%s: NHWC 16-bit view with 8-element aligned strides expected", what);
  return FD_OK;
}

void fill_pro(const FdPrologue* pro, int& mode, float& slope, float& eps, const float*& mean, const float*& var,
              const float*& gamma, const float*& beta) {
  mode = 0, slope = 1.f, eps = 1e-5f, mean = var = gamma = beta = nullptr;
  if (!pro) return;
  slope = pro->act == FD_ACT_RELU ? 0.f : (pro->act == FD_ACT_LEAKY02 ? 0.2f : 1.f);
  mode = pro->mean ? 2 : (pro->act != FD_ACT_NONE ? 1 : 0);
  eps = pro->eps;
  mean = pro->mean, var = pro->var, gamma = pro->gamma, beta = pro->beta;
}

}  // namespace

extern "C" int fdgan_conv2d_bwd_weight(const FdTensor* x, const FdPrologue* pro, const FdTensor* dy, const FdConvDesc* d,
                                       float* dw, float* dbias, float* workspace, int64_t workspace_floats,
                                       int accumulate, FdStream stream) {
  return fdgan_conv2d_bwd_weight_job(x, pro, dy, d, dw, dbias, workspace, workspace_floats, accumulate, nullptr, 0, stream);
}

extern "C" int fdgan_wgrad_tr_reduce_batch(const FdTrReduceJob* jobs_device, int64_t njobs, int64_t total_groups, FdStream stream) {
  FD_REQUIRE(jobs_device && njobs > 0 && njobs < (1 << 20) && total_groups > 0 && total_groups < (1ll << 31), "wgrad_tr_reduce_batch: empty job table");
  return conv_wgrad_tr_reduce_batch(jobs_device, njobs, total_groups, static_cast<hipStream_t>(stream));
}

extern "C" int fdgan_conv2d_bwd_weight_job(const FdTensor* x, const FdPrologue* pro, const FdTensor* dy, const FdConvDesc* d,
                                           float* dw, float* dbias, float* workspace, int64_t workspace_floats,
                                           int accumulate, FdTrReduceJob* job, int defer, FdStream stream) {
  if (job) *job = FdTrReduceJob{};
  if (int rc = check_view(x, "conv2d_bwd_weight(x)", FD_F16)) return rc;
  if (int rc = check_view(dy, "conv2d_bwd_weight(dy)")) return rc;
  FD_REQUIRE(d && dw, "conv2d_bwd_weight: NULL descriptor / dw");
  FD_REQUIRE(!d->upsample2, "conv2d_bwd_weight: pass the gradient of the PRE-upsample output (fdgan_grad_ew mode 2)");
  FD_REQUIRE(pro == nullptr || pro->act == FD_ACT_NONE || pro->act == FD_ACT_RELU || pro->act == FD_ACT_LEAKY02,
             "conv2d_bwd_weight: prologue activation %d", pro ? pro->act : 0);
  const bool pool = pro && pro->pool2;
  FD_REQUIRE(!pool || (d->ksize == 1 && d->stride == 1 && d->pad == 0), "conv2d_bwd_weight: pooled prologue needs a 1x1 conv");
  const long long hin = pool ? x->h / 2 : x->h, win = pool ? x->w / 2 : x->w;
  const long long ho = (hin + 2 * d->pad - d->ksize) / d->stride + 1, wo = (win + 2 * d->pad - d->ksize) / d->stride + 1;
  FD_REQUIRE(dy->n == x->n && dy->h == ho && dy->w == wo, "conv2d_bwd_weight: dy is %lldx%lld, expected %lldx%lld",
             (long long)dy->h, (long long)dy->w, ho, wo);
  const int cout = d->cout ? d->cout : (int)dy->c;
  WgradArgs a{};
  a.x = static_cast<const unsigned short*>(x->ptr);
  a.x_sn = x->stride[0], a.x_sh = (int)x->stride[1], a.x_sw = (int)x->stride[2];
  a.Hs = (int)x->h, a.Ws = (int)x->w, a.Cin = (int)x->c, a.Cin8 = (int)((x->c + 7) / 8);
  a.dy = static_cast<const unsigned short*>(dy->ptr);
  a.dy_sn = dy->stride[0], a.dy_sh = (int)dy->stride[1], a.dy_sw = (int)dy->stride[2];
  a.Ho = (int)ho, a.Wo = (int)wo, a.Cout = cout, a.Cout8 = (cout + 7) / 8;
  FD_REQUIRE(a.Cout8 * 8 <= dy->stride[2] && a.Cin8 * 8 <= x->stride[2], "conv2d_bwd_weight: channel padding exceeds the pixel pitch");
  a.ks = d->ksize, a.stride = d->stride, a.pad = d->pad;
  a.P = (long long)x->n * ho * wo;
  a.pool = pool ? 1 : 0;
  fill_pro(pro, a.pro_mode, a.p_slope, a.eps, a.p_mean, a.p_var, a.p_gamma, a.p_beta);
  if (pool && a.pro_mode == 0) a.pro_mode = 1;   // the pooled path always goes through the affine helper (scale 1, shift 0)
  // the discriminators' last conv (ONE filter): a vector-ALU streaming reduction (conv_c1.hip)
  if (cout == 1 && !pool && dbias == nullptr) {
    long long ns = 0;
    hipStream_t st1 = static_cast<hipStream_t>(stream);
    const int rc1 = wgrad_cout1_launch(x, dy, d->ksize, d->stride, d->pad, a.pro_mode, a.p_slope, a.eps, a.p_mean, a.p_var, a.p_gamma,
                                       a.p_beta, workspace, workspace_floats, &ns, st1);
    if (rc1 < 0) return rc1;
    if (rc1 == 0) {
      const long long numel1 = (long long)a.Cin * d->ksize * d->ksize;
      WredArgs r1{workspace, dw, numel1, (int)ns, accumulate};
      return fd_wgrad_reduce_args(r1, st1);
    }
  }
  // the few-channel convs at the ends of the networks (3 -> 64, 16 -> 3, 9 -> 36 stride 2): one GEMM over all taps
  if (workspace != nullptr && !pool && a.Cin <= 16 && cout <= 64) {
    long long ns = 0;
    hipStream_t sts = static_cast<hipStream_t>(stream);
    const int rcs = conv_wgrad_small_launch(x, dy, cout, d->ksize, d->stride, d->pad, a.pro_mode, a.p_slope, a.eps, a.p_mean, a.p_var,
                                            a.p_gamma, a.p_beta, dbias != nullptr, workspace, workspace_floats, &ns, sts);
    if (rcs < 0) return rcs;
    if (rcs == 0) {
      const long long numels = (long long)cout * a.Cin * d->ksize * d->ksize;
      WredArgs rs{workspace, dw, numels, (int)ns, accumulate};
      if (int rc = fd_wgrad_reduce_args(rs, sts)) return rc;
      if (dbias == nullptr) return FD_OK;
      WredArgs rb{workspace + ns * numels, dbias, cout, (int)ns, accumulate};
      return fd_wgrad_reduce_args(rb, sts);
    }
  }
  // row-walking transpose-read kernels (conv_wgrad_tr.hip): the growth conv and the discriminator's 4x4 conv
  const int trv = workspace != nullptr ? conv_wgrad_tr_variant(cout, a.Cin, d->ksize, d->stride, d->pad, pool) : 0;
  if (trv != 0) {
    WgradRowsArgs w{};
    w.x = a.x, w.x_sn = a.x_sn, w.x_sh = a.x_sh, w.x_sw = a.x_sw;
    w.dy = a.dy, w.dy_sn = a.dy_sn, w.dy_sh = a.dy_sh, w.dy_sw = a.dy_sw;
    w.H = a.Hs, w.W = a.Ws, w.Cin = a.Cin, w.Cout = cout, w.Ho = a.Ho, w.Wo = a.Wo, w.pad = a.pad;
    w.pro_mode = a.pro_mode, w.p_slope = a.p_slope, w.eps = a.eps;
    w.p_mean = a.p_mean, w.p_var = a.p_var, w.p_gamma = a.p_gamma, w.p_beta = a.p_beta;
    const int rc = conv_wgrad_tr_launch(trv, w, x->n, workspace, workspace_floats, dw, dbias, accumulate, static_cast<hipStream_t>(stream),
                                        job, defer);
    if (rc != 1) {
      if (job && dbias) job->part = nullptr;   // a bias gradient: nothing was deferred
      return rc;
    }
    if (job) *job = FdTrReduceJob{};           // 1: workspace too small for this kernel's partials
  }
  // the dense-layer growth conv: all nine taps in one workgroup
  if (workspace != nullptr && dbias == nullptr && d->ksize == 3 && d->stride == 1 && d->pad == 1 && !pool && cout <= 32 &&
      a.Cin % 32 == 0 && a.Ws % 4 == 0 && FD_TUNE_GETENV("FDGAN_DEBUG_NO_WGRAD3") == nullptr) {
    Wgrad3Args w{};
    w.x = a.x, w.x_sn = a.x_sn, w.x_sh = a.x_sh, w.x_sw = a.x_sw;
    w.dy = a.dy, w.dy_sn = a.dy_sn, w.dy_sh = a.dy_sh, w.dy_sw = a.dy_sw;
    w.H = a.Hs, w.W = a.Ws, w.Cin = a.Cin, w.Cout = cout, w.Cout8 = a.Cout8;
    w.xblocks = (a.Ws + W3_PB - 1) / W3_PB;
    w.pro_mode = a.pro_mode, w.p_slope = a.p_slope, w.eps = a.eps;
    w.p_mean = a.p_mean, w.p_var = a.p_var, w.p_gamma = a.p_gamma, w.p_beta = a.p_beta;
    const long long numel3 = (long long)cout * a.Cin * 9;
    const long long strips = (long long)x->n * w.xblocks, ci_tiles = a.Cin / 32;
    long long segs = 1024 / (strips * ci_tiles);                 // ~1024 workgroups in all
    if (segs < 1) segs = 1;
    if (segs > (a.Hs + 7) / 8) segs = (a.Hs + 7) / 8;            // at least 8 rows per item (2 rows of halo re-staged per item)
    while (segs > 1 && strips * segs * numel3 > workspace_floats) --segs;
    if (strips * segs * numel3 <= workspace_floats && strips * segs < 65536) {
      w.segs = (int)segs;
      w.seg_rows = (int)((a.Hs + segs - 1) / segs);
      w.part = workspace;
      static bool attr_done = false;
      if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad3x3_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) FD_FAIL(FD_ELAUNCH, "hipFuncSetAttribute(conv_wgrad3x3): %s", hipGetErrorString(e));
        attr_done = true;
      }
      hipStream_t st3 = static_cast<hipStream_t>(stream);
      int rc3 = fd_launch(&conv_wgrad3x3_kernel, "conv_wgrad3x3", dim3((unsigned)ci_tiles, (unsigned)(strips * segs)), dim3(256),
                          W3_XS_B + W3_DT_B + 256, w, st3);
      if (rc3 != FD_OK) return rc3;
      WredArgs r3{workspace, dw, numel3, (int)(strips * segs), accumulate};
      return fd_wgrad_reduce_args(r3, st3);
    }
  }
  // the dense-layer bottleneck (1x1, 128 filters): transpose-read kernel (conv_wgrad1x1_tr.hip)
  if (workspace != nullptr && conv_wgrad1x1_tr_fits(x, dy, cout, d->ksize, d->stride, pool, dbias != nullptr)) {
    long long ns = 0;
    hipStream_t st1 = static_cast<hipStream_t>(stream);
    const int rc1 = conv_wgrad1x1_tr_launch(x, dy, a.pro_mode, a.p_slope, a.eps, a.p_mean, a.p_var, a.p_gamma, a.p_beta, workspace,
                                            workspace_floats, &ns, st1);
    if (rc1 < 0) return rc1;
    if (rc1 == FD_OK) {
      const long long numel1 = 128LL * a.Cin;
      WredArgs r1{workspace, dw, numel1, (int)ns, accumulate};
      return fd_wgrad_reduce_args(r1, st1);
    }
  }
  const long long numel = (long long)cout * a.Cin * d->ksize * d->ksize;
  // workgroup tile: 64 x 64 with 4 waves, or 128 x 128 with 8 waves (half the L2 -> LDS traffic per flop: the 64-tile
  // kernel runs at the ~5 TB/s its operand re-reads can be served at) when both channel counts fill it
  static const char* tsel = FD_TUNE_GETENV("FDGAN_DEBUG_WGRAD_T");   // tuning aid: force 64 / 128
  const bool fits128 = cout >= 96 && a.Cin >= 96 && (cout % 128 == 0 || cout % 128 > 64) && (a.Cin % 128 == 0 || a.Cin % 128 > 32);
  const int T = tsel ? atoi(tsel) : (fits128 ? 128 : 64);
  const long long base = (long long)((a.Cin + T - 1) / T) * ((cout + T - 1) / T) * d->ksize * d->ksize;
  // split the pixel axis until ~768 (T = 64: 3-4 resident per CU) / ~512 (T = 128: 2 per CU) workgroups exist
  long long nsplit = 1;
  if (workspace != nullptr) {
    nsplit = (T == 128 ? 512 : 768) / base;
    static const char* minpx_env = FD_TUNE_GETENV("FDGAN_DEBUG_WGRAD_MINPX");   // tuning aid
    const long long minpx = minpx_env ? atoll(minpx_env) : 2048;
    long long max_by_px = (a.P + minpx - 1) / minpx;
    // few pixels AND few tiles (the transitions at 32x32, the narrow 1x1s): 2048 pixels per split would leave most CUs without a
    // workgroup -- 256 -> 128 pooled at 32x32 ran as 16 workgroups, 105 us.  Split down to 256 pixels then, up to one round of
    // workgroups (T = 128: one per CU; T = 64: two): 105 -> 29 us, 64 -> 32 pooled 74 -> 58, 512 -> 64 40 -> 31, 96 -> 16 44 -> 35
    const long long one_round = T == 128 ? 256 : 512;
    if (!minpx_env && base * max_by_px < one_round) {
      max_by_px = (a.P + 255) / 256;
      const long long fill = one_round / base > 0 ? one_round / base : 1;
      if (nsplit > fill) nsplit = fill;
    }
    if (nsplit > max_by_px) nsplit = max_by_px;
    const long long per = numel + (dbias ? cout : 0);
    if (nsplit * per > workspace_floats) nsplit = workspace_floats / per;
    if (nsplit < 1) nsplit = 1;
  }
  FD_REQUIRE(workspace != nullptr || !accumulate, "conv2d_bwd_weight: accumulate needs a workspace");
  const bool direct = workspace == nullptr;
  FD_REQUIRE(direct || numel + (dbias ? cout : 0) <= workspace_floats, "conv2d_bwd_weight: workspace too small (%lld floats)", numel + cout);
  static const char* ph = FD_TUNE_GETENV("FDGAN_DEBUG_PHASES");
  a.dbg_skip = ph ? atoi(ph) : 0;
  a.nsplit = (int)nsplit;
  a.split_px = ((a.P + nsplit - 1) / nsplit + WG_KPX - 1) / WG_KPX * WG_KPX;
  a.dw = direct ? dw : workspace;
  a.dbias = dbias ? (direct ? dbias : workspace + nsplit * numel) : nullptr;
  a.tiles_ci = (a.Cin + T - 1) / T, a.tiles_co = (cout + T - 1) / T;
  dim3 grid((unsigned)(base * nsplit));
  hipStream_t st = static_cast<hipStream_t>(stream);
  const unsigned lds = 2u * T * WG_ROWB + 2u * T * 4 + (dbias ? 32u * T * 4 : 0u);
  static bool attr128 = false;
  if (T == 128 && !attr128) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_kernel<128, 8, false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_kernel<128, 8, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
    if (e != hipSuccess) FD_FAIL(FD_ELAUNCH, "hipFuncSetAttribute(conv_wgrad<128>): %s", hipGetErrorString(e));
    attr128 = true;
  }
  int rc;
  if (a.pool) rc = T == 128 ? fd_launch(&conv_wgrad_kernel<128, 8, true>, "conv_wgrad_t128_pool", grid, dim3(512), lds, a, st)
                            : fd_launch(&conv_wgrad_kernel<64, 4, true>, "conv_wgrad_pool", grid, dim3(256), lds, a, st);
  else rc = T == 128 ? fd_launch(&conv_wgrad_kernel<128, 8, false>, "conv_wgrad_t128", grid, dim3(512), lds, a, st)
                     : fd_launch(&conv_wgrad_kernel<64, 4, false>, "conv_wgrad", grid, dim3(256), lds, a, st);
  if (rc != FD_OK || direct) return rc;
  WredArgs r{workspace, dw, numel, (int)nsplit, accumulate};
  rc = fd_wgrad_reduce_args(r, st);
  if (rc != FD_OK || !dbias) return rc;
  WredArgs rb{workspace + nsplit * numel, dbias, cout, (int)nsplit, accumulate};
  return fd_wgrad_reduce_args(rb, st);
}

extern "C" int fdgan_bn_act_bwd(const FdTensor* da, const FdTensor* x, const FdPrologue* pro, float* partial,
                                int64_t capacity_floats, int64_t* rows_out, int64_t* cpad_out, FdStream stream) {
  return fdgan_bn_act_bwd_acc(da, x, pro, nullptr, partial, capacity_floats, rows_out, cpad_out, stream);
}

extern "C" int fdgan_bn_act_bwd_acc(const FdTensor* da, const FdTensor* x, const FdPrologue* pro, const FdTensor* dx, float* partial,
                                    int64_t capacity_floats, int64_t* rows_out, int64_t* cpad_out, FdStream stream) {
  return fdgan_bn_act_bwd_dx(da, x, pro, dx, 0, partial, capacity_floats, rows_out, cpad_out, stream);
}

extern "C" int fdgan_bn_act_bwd_dx(const FdTensor* da, const FdTensor* x, const FdPrologue* pro, const FdTensor* dx, int dx_store,
                                   float* partial, int64_t capacity_floats, int64_t* rows_out, int64_t* cpad_out, FdStream stream) {
  if (int rc = check_view(da, "bn_act_bwd(da)")) return rc;
  if (int rc = check_view(x, "bn_act_bwd(x)", FD_F16)) return rc;
  const bool pooled = pro && pro->pool2;
  FD_REQUIRE(da->n == x->n && da->c == x->c && (pooled ? (da->h == x->h / 2 && da->w == x->w / 2 && x->h % 2 == 0 && x->w % 2 == 0)
                                                       : (da->h == x->h && da->w == x->w)),
             "bn_act_bwd: da / x shape mismatch");
  FD_REQUIRE(!pooled || (pro->mean && partial), "bn_act_bwd: the pooled form only produces the BatchNorm sums");
  BnActBwdArgs a{};
  a.pool = pooled ? 1 : 0;
  if (dx != nullptr) {
    if (int rc = check_view(dx, "bn_act_bwd(dx)")) return rc;
    FD_REQUIRE(pooled && dx->n == x->n && dx->h == x->h && dx->w == x->w && dx->c == x->c, "bn_act_bwd: dx goes with the pooled form and has x's shape");
    a.dx = static_cast<unsigned short*>(dx->ptr), a.dx_sn = dx->stride[0], a.dx_sh = (int)dx->stride[1], a.dx_sw = (int)dx->stride[2];
  }
  a.da = static_cast<unsigned short*>(da->ptr);
  a.da_sn = da->stride[0], a.da_sh = (int)da->stride[1], a.da_sw = (int)da->stride[2];
  a.x = static_cast<const unsigned short*>(x->ptr);
  a.x_sn = x->stride[0], a.x_sh = (int)x->stride[1], a.x_sw = (int)x->stride[2];
  a.H = (int)x->h, a.W = (int)x->w, a.C = (int)x->c, a.C8 = (int)((x->c + 7) / 8);
  a.P = (long long)x->n * x->h * x->w;
  FD_REQUIRE(a.P < (1ll << 31), "bn_act_bwd: more than 2^31 pixels");
  fill_pro(pro, a.pro_mode, a.slope, a.eps, a.mean, a.var, a.gamma, a.beta);
  a.cpad = a.C8 * 8;
  long long rows = (a.P + 31) / 32;
  static const char* cap_env = FD_TUNE_GETENV("FDGAN_DEBUG_BN_ROWS");   // experiment aid
  const long long chunks = (a.C8 + 7) / 8;                      // 64-channel chunks -> gridDim.y
  long long cap = (cap_env ? atoll(cap_env) : 512) / chunks;    // ~2 workgroups per CU in total measured best
  if (cap < 16) cap = 16;
  if (rows > cap) rows = cap;
  a.partial = (a.pro_mode == 2) ? partial : nullptr;
  if (a.partial) FD_REQUIRE(rows * a.cpad * 2 <= capacity_floats, "bn_act_bwd: workspace too small (%lld floats needed)", rows * a.cpad * 2);
  if (rows_out) *rows_out = rows;
  if (cpad_out) *cpad_out = a.cpad;
  const dim3 grid((unsigned)rows, (unsigned)chunks);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (a.pool) {
    if (a.dx && dx_store) return fd_launch(&bn_act_bwd_kernel<true, 2>, "bn_act_bwd", grid, dim3(256), 0, a, st);
    return a.dx ? fd_launch(&bn_act_bwd_kernel<true, 1>, "bn_act_bwd", grid, dim3(256), 0, a, st)
                : fd_launch(&bn_act_bwd_kernel<true, 0>, "bn_act_bwd", grid, dim3(256), 0, a, st);
  }
  return a.dx ? fd_launch(&bn_act_bwd_kernel<false, 1>, "bn_act_bwd", grid, dim3(256), 0, a, st)
              : fd_launch(&bn_act_bwd_kernel<false, 0>, "bn_act_bwd", grid, dim3(256), 0, a, st);
}

extern "C" int fdgan_bn_bwd_finalize(const float* partial, int64_t rows, int64_t cpad, int64_t channels, float* dgamma,
                                     float* dbeta, int accumulate, FdStream stream) {
  return fdgan_bn_bwd_finalize_sink(partial, rows, cpad, channels, dgamma, dbeta, accumulate, nullptr, nullptr, stream);
}

static int launch_sum_finalize(SumFinArgs a, float* scratch, int64_t scratch_floats, FdStream stream) {
  {  // measurement aid (tuning builds; results wrong): after N finalize launches skip them all -- what do these single-workgroup launches
     // cost the step IN PLACE, queued behind the side stream's workgroups at every boundary?  (upper bound for any scheme that removes them)
    static const char* skip = FD_TUNE_GETENV("FDGAN_DEBUG_SKIP_FINALIZE");
    static long long calls = 0;
    if (skip && ++calls > atoll(skip)) return FD_OK;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 cgrid((unsigned)((a.channels + 31) / 32));
  constexpr int SLICES = 32;
  if (scratch != nullptr && a.rows > 256 && (int64_t)SLICES * a.cpad * 2 <= scratch_floats) {
    SumFinArgs l1 = a;
    l1.level1 = scratch;
    if (int rc = fd_launch(&sum_finalize_kernel, "bn_bwd_finalize_l1", dim3(cgrid.x, SLICES), dim3(1024), 0, l1, st)) return rc;
    a.partial = scratch, a.rows = SLICES;
  }
  a.level1 = nullptr;
  return fd_launch(&sum_finalize_kernel, "bn_bwd_finalize", cgrid, dim3(1024), 0, a, st);
}

extern "C" int fdgan_bn_bwd_finalize_sink(const float* partial, int64_t rows, int64_t cpad, int64_t channels, float* dgamma,
                                          float* dbeta, int accumulate, float* sink_dgamma, float* sink_dbeta, FdStream stream) {
  FD_REQUIRE(partial && dgamma && dbeta && rows > 0 && channels > 0 && cpad >= channels, "bn_bwd_finalize: bad arguments");
  SumFinArgs a{partial, rows, cpad, channels, dbeta, dgamma, accumulate, sink_dbeta, sink_dgamma, nullptr, nullptr, 0.f, nullptr,
               nullptr, 0.f, nullptr, nullptr};
  return launch_sum_finalize(a, nullptr, 0, stream);
}

extern "C" int fdgan_bn_bwd_finalize_raw(const float* partial, int64_t rows, int64_t cpad, int64_t channels, const float* mean,
                                         const float* var, float eps, float* dgamma, float* dbeta, float* sink_dgamma,
                                         float* sink_dbeta, float* scratch, int64_t scratch_floats, FdStream stream) {
  FD_REQUIRE(partial && dgamma && dbeta && mean && var && rows > 0 && channels > 0 && cpad >= channels, "bn_bwd_finalize_raw: bad arguments");
  SumFinArgs a{partial, rows, cpad, channels, dbeta, dgamma, 0, sink_dbeta, sink_dgamma, mean, var, eps, nullptr,
               nullptr, 0.f, nullptr, nullptr};
  return launch_sum_finalize(a, scratch, scratch_floats, stream);
}

/* fdgan_bn_bwd_finalize_raw + fdgan_bn_bwd_coef in one launch: the reduced (dbeta, dgamma) go to the parameters' gradient
 * sinks (optional) and straight into the buffer's deferred coefficient pair; they are not stored anywhere else. */
extern "C" int fdgan_bn_bwd_finalize_coef(const float* partial, int64_t rows, int64_t cpad, int64_t channels, const FdPrologue* pro,
                                          int64_t count, float* sink_dgamma, float* sink_dbeta, float* bsum, float* csum,
                                          float* scratch, int64_t scratch_floats, int coef_store, FdStream stream) {
  FD_REQUIRE(partial && pro && pro->mean && pro->var && bsum && csum && rows > 0 && channels > 0 && cpad >= channels && count > 0,
             "bn_bwd_finalize_coef: bad arguments");
  SumFinArgs a{partial, rows, cpad, channels, nullptr, nullptr, 0, sink_dbeta, sink_dgamma, pro->mean, pro->var, pro->eps, nullptr,
               pro->gamma, 1.f / (float)count, bsum, csum, coef_store};
  return launch_sum_finalize(a, scratch, scratch_floats, stream);
}

// ---- deferred affine part of BatchNorm's backward ---------------------------------------------------------------
// dx = A * dpre + B * x + C (see bn_bwd_apply).  fdgan_conv2d_bwd_data(accumulate = 1) already added A * dpre to the
// gradient buffer from its epilogue; B and C depend on the reductions, are per channel, and are LINEAR in x -- so the
// layers of a dense block that normalise the same channels add their (B, C) into one coefficient pair and a single
// pass dx += Bsum * x + Csum serves them all, right before the gradient of those channels is consumed.
struct BnCoefArgs {
  const float *dgamma, *dbeta, *mean, *var, *gamma;
  float eps, inv_m;
  int channels;
  float *bsum, *csum;
};
__global__ __launch_bounds__(256) void bn_bwd_coef_kernel(BnCoefArgs a) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= a.channels) return;
  const float rs = 1.f / sqrtf(a.var[c] + a.eps), gmm = a.gamma ? a.gamma[c] : 1.f;
  const float A = gmm * rs, B = -gmm * rs * rs * a.dgamma[c] * a.inv_m;
  a.bsum[c] += B;
  a.csum[c] += -A * a.dbeta[c] * a.inv_m - B * a.mean[c];
}

struct AffineAccArgs {
  const unsigned short* x;
  long long x_sn;
  int x_sh, x_sw;
  unsigned short* dx;
  long long dx_sn;
  int dx_sh, dx_sw;
  int H, W, C, C8;
  long long P;
  const float *bsum, *csum;
  int gpp, dense;   // channel groups per pixel in a workgroup pass (<= 8); both views pixel-dense
  unsigned short* out;   // where dx + B x + C goes: dx itself (in place), or a separate view (fdgan_affine_accumulate_out)
  long long o_sn;
  int o_sh, o_sw;
};
template <bool DENSE>
__global__ __launch_bounds__(256) void affine_acc_kernel(AffineAccArgs a) {
  // G channel groups x (256 / G) pixels per workgroup pass: a 32-channel slice (G = 4) keeps all 256 threads busy (the fixed
  // 8 x 32 split left half of them idle); pixel offsets are p * pitch on pixel-dense views (every concat buffer is), 32-bit
  // (image, row, column) arithmetic otherwise -- the 64-bit divisions of the first version were most of its instructions
  const int G = a.gpp, ppp = 256 / G;
  const int slot = (int)threadIdx.x / G, grp = (int)threadIdx.x - slot * G;
  const int c8 = blockIdx.y * G + grp;
  if (slot >= ppp || c8 >= a.C8) return;
  const unsigned HWu = (unsigned)(a.H * a.W), Wu = (unsigned)a.W;
  // No branch around a load anywhere below: out-of-range lanes read a clamped (valid) address and only their STORE is predicated.
  // With `if (p < a.P) load` hipcc put s_waitcnt vmcnt(0) at every join, i.e. right behind each pair of loads -- of the eight
  // 16-byte loads a thread issues per pass at most four were ever in flight (2.2 TB/s in the step; same finding as
  // csrc/freqsplit.hip's row fetch, round 4).
  float B[8], Cc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = c8 * 8 + e, cc = min(c, a.C - 1);
    const float b = a.bsum[cc], k = a.csum[cc];
    B[e] = c < a.C ? b : 0.f;
    Cc[e] = c < a.C ? k : 0.f;
  }
  const long long stride = (long long)gridDim.x * ppp;
  for (long long p0 = (long long)blockIdx.x * ppp + slot; p0 < a.P; p0 += 4 * stride) {
    u32x4 xv[4], gv[4];
    unsigned short* op[4];
    bool ok[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const long long pk = p0 + k * stride;
      ok[k] = pk < a.P;
      const long long p = ok[k] ? pk : p0;
      long long xo, go, oo;
      if constexpr (DENSE) {
        xo = p * a.x_sw, go = p * a.dx_sw, oo = p * a.o_sw;
      } else {
        const unsigned pu = (unsigned)p, n = pu / HWu, r = pu - n * HWu;
        const int y = (int)(r / Wu), xx = (int)(r - (unsigned)y * Wu);
        xo = n * a.x_sn + (long long)y * a.x_sh + (long long)xx * a.x_sw;
        go = n * a.dx_sn + (long long)y * a.dx_sh + (long long)xx * a.dx_sw;
        oo = n * a.o_sn + (long long)y * a.o_sh + (long long)xx * a.o_sw;
      }
      xv[k] = *reinterpret_cast<const u32x4*>(a.x + xo + c8 * 8);
      op[k] = a.out + oo + c8 * 8;
      gv[k] = *reinterpret_cast<const u32x4*>(a.dx + go + c8 * 8);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const f32x8 xf = fd_cvt8<FmtA>(xv[k]);      // the normalised tensor: a forward activation, fp16
      f32x8 o = __builtin_convertvector(__builtin_bit_cast(bf16x8, gv[k]), f32x8);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] += fmaf(B[e], xf[e], Cc[e]);
      // a small coherent term on top of a value already on the bf16 grid: stochastic rounding (csrc/common.h: fd_pk8_sr)
      const u32x4 res = fd_pk8_sr(o, (unsigned)((p0 + k * stride) * (long long)a.C8) + (unsigned)c8);
      if (ok[k]) *reinterpret_cast<u32x4*>(op[k]) = res;
    }
  }
}

extern "C" int fdgan_bn_bwd_coef(const float* dgamma, const float* dbeta, const FdPrologue* pro, int64_t channels, int64_t count,
                                 float* bsum, float* csum, FdStream stream) {
  FD_REQUIRE(dgamma && dbeta && pro && pro->mean && pro->var && bsum && csum && channels > 0 && count > 0, "bn_bwd_coef: bad arguments");
  BnCoefArgs a{dgamma, dbeta, pro->mean, pro->var, pro->gamma, pro->eps, 1.f / (float)count, (int)channels, bsum, csum};
  return fd_launch(&bn_bwd_coef_kernel, "bn_bwd_coef", dim3((unsigned)((channels + 255) / 256)), dim3(256), 0, a,
                   static_cast<hipStream_t>(stream));
}

static int affine_accumulate_impl(const FdTensor* x, const float* bsum, const float* csum, const FdTensor* dx, const FdTensor* out, FdStream stream);
extern "C" int fdgan_affine_accumulate(const FdTensor* x, const float* bsum, const float* csum, const FdTensor* dx, FdStream stream) {
  return affine_accumulate_impl(x, bsum, csum, dx, dx, stream);
}
extern "C" int fdgan_affine_accumulate_out(const FdTensor* x, const float* bsum, const float* csum, const FdTensor* g, const FdTensor* out,
                                          FdStream stream) {
  FD_REQUIRE(out != nullptr, "affine_accumulate_out: NULL output view");
  return affine_accumulate_impl(x, bsum, csum, g, out, stream);
}
static int affine_accumulate_impl(const FdTensor* x, const float* bsum, const float* csum, const FdTensor* dx, const FdTensor* out, FdStream stream) {
  if (int rc = check_view(x, "affine_accumulate(x)", FD_F16)) return rc;
  if (int rc = check_view(dx, "affine_accumulate(dx)")) return rc;
  if (int rc = check_view(out, "affine_accumulate(out)")) return rc;
  FD_REQUIRE(bsum && csum && dx->n == x->n && dx->h == x->h && dx->w == x->w && dx->c == x->c, "affine_accumulate: shape mismatch");
  FD_REQUIRE(out->n == x->n && out->h == x->h && out->w == x->w && out->c == x->c, "affine_accumulate: output shape mismatch");
  AffineAccArgs a{};
  a.out = static_cast<unsigned short*>(out->ptr);
  a.o_sn = out->stride[0], a.o_sh = (int)out->stride[1], a.o_sw = (int)out->stride[2];
  a.x = static_cast<const unsigned short*>(x->ptr);
  a.x_sn = x->stride[0], a.x_sh = (int)x->stride[1], a.x_sw = (int)x->stride[2];
  a.dx = static_cast<unsigned short*>(dx->ptr);
  a.dx_sn = dx->stride[0], a.dx_sh = (int)dx->stride[1], a.dx_sw = (int)dx->stride[2];
  a.H = (int)x->h, a.W = (int)x->w, a.C = (int)x->c, a.C8 = (int)((x->c + 7) / 8);
  a.P = (long long)x->n * x->h * x->w;
  a.bsum = bsum, a.csum = csum;
  FD_REQUIRE(a.P < (1ll << 31), "affine_accumulate: more than 2^31 pixels");
  auto pixel_dense = [](const FdTensor* t) { return t->stride[1] == t->w * t->stride[2] && t->stride[0] == t->h * t->stride[1]; };
  a.dense = pixel_dense(x) && pixel_dense(dx) && pixel_dense(out) ? 1 : 0;
  a.gpp = a.C8 < 8 ? a.C8 : 8;
  const int ppp = 256 / a.gpp;
  const long long chunks = (a.C8 + a.gpp - 1) / a.gpp;
  long long rows = (a.P + ppp - 1) / ppp, cap = 1024 / chunks;      // ~4 workgroups per CU in all
  if (cap < 16) cap = 16;
  if (rows > cap) rows = cap;
  if (a.dense)
    return fd_launch(&affine_acc_kernel<true>, "affine_accumulate", dim3((unsigned)rows, (unsigned)chunks), dim3(256), 0, a,
                     static_cast<hipStream_t>(stream));
  return fd_launch(&affine_acc_kernel<false>, "affine_accumulate", dim3((unsigned)rows, (unsigned)chunks), dim3(256), 0, a,
                   static_cast<hipStream_t>(stream));
}

extern "C" int fdgan_bn_bwd_apply(const FdTensor* dpre, const FdTensor* x, const FdPrologue* pro, const float* dgamma,
                                  const float* dbeta, const FdTensor* dx, int accumulate, FdStream stream) {
  if (int rc = check_view(dpre, "bn_bwd_apply(dpre)")) return rc;
  if (int rc = check_view(x, "bn_bwd_apply(x)", FD_F16)) return rc;
  if (int rc = check_view(dx, "bn_bwd_apply(dx)")) return rc;
  FD_REQUIRE(pro && pro->mean && pro->var && dgamma && dbeta, "bn_bwd_apply: needs the forward batch statistics and dgamma/dbeta");
  const bool pooled = pro->pool2 != 0;
  FD_REQUIRE(dpre->n == x->n && dpre->c == x->c && dx->n == x->n && dx->h == x->h && dx->w == x->w && dx->c == x->c &&
                 (pooled ? (dpre->h == x->h / 2 && dpre->w == x->w / 2 && x->h % 2 == 0 && x->w % 2 == 0)
                         : (dpre->h == x->h && dpre->w == x->w)),
             "bn_bwd_apply: shape mismatch");
  FD_REQUIRE(pro->act == FD_ACT_NONE || pro->act == FD_ACT_RELU || pro->act == FD_ACT_LEAKY02, "bn_bwd_apply: prologue activation %d", pro->act);
  BnApplyArgs a{};
  a.pool = pooled ? 1 : 0;
  a.beta = pro->beta;
  a.slope = pro->act == FD_ACT_RELU ? 0.f : (pro->act == FD_ACT_LEAKY02 ? 0.2f : 1.f);
  a.dpre = static_cast<const unsigned short*>(dpre->ptr);
  a.dp_sn = dpre->stride[0], a.dp_sh = (int)dpre->stride[1], a.dp_sw = (int)dpre->stride[2];
  a.x = static_cast<const unsigned short*>(x->ptr);
  a.x_sn = x->stride[0], a.x_sh = (int)x->stride[1], a.x_sw = (int)x->stride[2];
  a.dx = static_cast<unsigned short*>(dx->ptr);
  a.dx_sn = dx->stride[0], a.dx_sh = (int)dx->stride[1], a.dx_sw = (int)dx->stride[2];
  a.H = (int)x->h, a.W = (int)x->w, a.C = (int)x->c, a.C8 = (int)((x->c + 7) / 8);
  a.P = (long long)x->n * x->h * x->w;
  FD_REQUIRE(a.P < (1ll << 31), "bn_bwd_apply: more than 2^31 pixels");
  a.eps = pro->eps;
  a.inv_m = 1.f / (float)a.P;
  a.mean = pro->mean, a.var = pro->var, a.gamma = pro->gamma, a.dbeta = dbeta, a.dgamma = dgamma;
  a.accumulate = accumulate;
  long long rows = (a.P + 31) / 32;
  static const char* cap_env = FD_TUNE_GETENV("FDGAN_DEBUG_BN_ROWS");   // experiment aid
  const long long chunks = (a.C8 + 7) / 8;
  long long cap = (cap_env ? atoll(cap_env) : 512) / chunks;
  if (cap < 16) cap = 16;
  if (rows > cap) rows = cap;
  const dim3 grid((unsigned)rows, (unsigned)chunks);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (a.pool) return a.accumulate ? fd_launch(&bn_bwd_apply_kernel<true, true>, "bn_bwd_apply", grid, dim3(256), 0, a, st)
                                  : fd_launch(&bn_bwd_apply_kernel<true, false>, "bn_bwd_apply", grid, dim3(256), 0, a, st);
  return a.accumulate ? fd_launch(&bn_bwd_apply_kernel<false, true>, "bn_bwd_apply", grid, dim3(256), 0, a, st)
                      : fd_launch(&bn_bwd_apply_kernel<false, false>, "bn_bwd_apply", grid, dim3(256), 0, a, st);
}

extern "C" int fdgan_conv2d_bwd_data_direct(const FdTensor* dy, const float* w, int cout, int cin, const FdConvDesc* d,
                                            float* dx, int64_t n, int64_t h, int64_t wd, FdStream stream) {
  if (int rc = check_view(dy, "conv2d_bwd_data_direct(dy)")) return rc;
  FD_REQUIRE(w && dx && d && cout > 0 && cin > 0, "conv2d_bwd_data_direct: bad arguments");
  const long long ho = (h + 2 * d->pad - d->ksize) / d->stride + 1, wo = (wd + 2 * d->pad - d->ksize) / d->stride + 1;
  FD_REQUIRE(dy->n == n && dy->h == ho && dy->w == wo && dy->c >= cout, "conv2d_bwd_data_direct: dy shape mismatch");
  DgradDirectArgs a{};
  a.dy = static_cast<const unsigned short*>(dy->ptr);
  a.dy_sn = dy->stride[0], a.dy_sh = (int)dy->stride[1], a.dy_sw = (int)dy->stride[2];
  a.Ho = (int)ho, a.Wo = (int)wo, a.Cout = cout;
  a.w = w, a.dx = dx;
  a.N = (int)n, a.Cin = cin, a.H = (int)h, a.W = (int)wd, a.ks = d->ksize, a.stride = d->stride, a.pad = d->pad;
  const unsigned lds_px = (unsigned)(d->ksize * d->ksize * cin * ((cout + 7) / 8 * 8) * 4);
  if ((cin == 3 || cin == 9 || cin == 16) && lds_px <= 60 * 1024 && (cout + 7) / 8 * 8 <= dy->stride[2] &&
      FD_TUNE_GETENV("FDGAN_DEBUG_NO_DGRAD_PX") == nullptr) {
    const dim3 grid((unsigned)((n * h * wd + 255) / 256));
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (cin == 3) return fd_launch(&dgrad_direct_px_kernel<3>, "dgrad_direct_px3", grid, dim3(256), lds_px, a, st);
    if (cin == 9) return fd_launch(&dgrad_direct_px_kernel<9>, "dgrad_direct_px9", grid, dim3(256), lds_px, a, st);
    return fd_launch(&dgrad_direct_px_kernel<16>, "dgrad_direct_px16", grid, dim3(256), lds_px, a, st);
  }
  const long long total = n * cin * h * wd;
  return fd_launch(&dgrad_direct_kernel, "dgrad_direct", dim3((unsigned)((total + 255) / 256)), dim3(256), 0, a,
                   static_cast<hipStream_t>(stream));
}

extern "C" int fdgan_out_act_bwd(const float* dout, const float* out, int64_t n, int64_t c, int64_t h, int64_t w, int act,
                                 const FdTensor* g, FdStream stream) {
  if (int rc = check_view(g, "out_act_bwd(g)")) return rc;
  FD_REQUIRE(dout && out && g->n == n && g->h == h && g->w == w && g->c >= c, "out_act_bwd: shape mismatch");
  FD_REQUIRE(act == FD_ACT_SIGMOID || act == FD_ACT_TANH || act == FD_ACT_NONE, "out_act_bwd: activation %d", act);
  OutActBwdArgs a{dout, out, static_cast<unsigned short*>(g->ptr), g->stride[0], (int)g->stride[1], (int)g->stride[2],
                  (int)n, (int)c, (int)h, (int)w, act, (int)((g->c + 7) / 8)};
  const long long total = n * h * w * a.C8;
  FD_REQUIRE(total < (1ll << 31), "out_act_bwd: more than 2^31 pieces");
  return fd_launch(&out_act_bwd_kernel, "out_act_bwd", dim3((unsigned)((total + 255) / 256)), dim3(256), 0, a,
                   static_cast<hipStream_t>(stream));
}

extern "C" int fdgan_grad_ew(int mode, const FdTensor* src, const FdTensor* ref, const FdTensor* dst, FdStream stream) {
  if (int rc = check_view(src, "grad_ew(src)")) return rc;
  if (int rc = check_view(dst, "grad_ew(dst)")) return rc;
  FD_REQUIRE(mode >= 0 && mode <= 4, "grad_ew: mode %d", mode);
  FD_REQUIRE(src->n == dst->n && src->c == dst->c, "grad_ew: batch / channel mismatch");
  if (mode == 0 || mode >= 3) FD_REQUIRE(src->h == dst->h && src->w == dst->w, "grad_ew: shape mismatch");
  if (mode == 1) FD_REQUIRE(src->h == dst->h / 2 && src->w == dst->w / 2, "grad_ew(unpool): src must be half of dst");
  if (mode == 2) FD_REQUIRE(src->h == dst->h * 2 && src->w == dst->w * 2, "grad_ew(sum-pool): src must be twice dst");
  GradEwArgs a{};
  a.src = static_cast<const unsigned short*>(src->ptr);
  a.s_sn = src->stride[0], a.s_sh = (int)src->stride[1], a.s_sw = (int)src->stride[2];
  if (mode >= 3) {
    if (int rc = check_view(ref, "grad_ew(ref)", FD_F16)) return rc;
    FD_REQUIRE(ref->n == dst->n && ref->h == dst->h && ref->w == dst->w && ref->c == dst->c, "grad_ew: ref shape mismatch");
    a.ref = static_cast<const unsigned short*>(ref->ptr);
    a.r_sn = ref->stride[0], a.r_sh = (int)ref->stride[1], a.r_sw = (int)ref->stride[2];
  }
  a.dst = static_cast<unsigned short*>(dst->ptr);
  a.d_sn = dst->stride[0], a.d_sh = (int)dst->stride[1], a.d_sw = (int)dst->stride[2];
  a.N = (int)dst->n, a.H = (int)dst->h, a.W = (int)dst->w, a.C8 = (int)((dst->c + 7) / 8), a.mode = mode, a.C = (int)dst->c;
  const long long total = (long long)a.N * a.H * a.W * a.C8;
  FD_REQUIRE(total < (1ll << 31), "grad_ew: more than 2^31 pieces");
  return fd_launch(&grad_ew_kernel, "grad_ew", dim3((unsigned)((total + 255) / 256)), dim3(256), 0, a,
                   static_cast<hipStream_t>(stream));
}

/* the same any-stride data gradient into an NHWC bf16 view (strided convolutions inside a plan: dehaze22.D's 4x4 s2) */
extern "C" int fdgan_conv2d_bwd_data_direct_nhwc(const FdTensor* dy, const float* w, int cout, int cin, const FdConvDesc* d,
                                                 const FdTensor* dx, FdStream stream) {
  if (int rc = check_view(dy, "conv2d_bwd_data_direct_nhwc(dy)")) return rc;
  if (int rc = check_view(dx, "conv2d_bwd_data_direct_nhwc(dx)")) return rc;
  FD_REQUIRE(w && d && cout > 0 && cin > 0 && dx->c >= cin, "conv2d_bwd_data_direct_nhwc: bad arguments");
  const long long ho = (dx->h + 2 * d->pad - d->ksize) / d->stride + 1, wo = (dx->w + 2 * d->pad - d->ksize) / d->stride + 1;
  FD_REQUIRE(dy->n == dx->n && dy->h == ho && dy->w == wo && dy->c >= cout, "conv2d_bwd_data_direct_nhwc: dy shape mismatch");
  DgradDirectArgs a{};
  a.dy = static_cast<const unsigned short*>(dy->ptr);
  a.dy_sn = dy->stride[0], a.dy_sh = (int)dy->stride[1], a.dy_sw = (int)dy->stride[2];
  a.Ho = (int)ho, a.Wo = (int)wo, a.Cout = cout;
  a.w = w, a.dx = nullptr;
  a.dx_nhwc = static_cast<unsigned short*>(dx->ptr);
  a.dxn_sn = dx->stride[0], a.dxn_sh = (int)dx->stride[1], a.dxn_sw = (int)dx->stride[2];
  a.N = (int)dx->n, a.Cin = cin, a.H = (int)dx->h, a.W = (int)dx->w, a.ks = d->ksize, a.stride = d->stride, a.pad = d->pad;
  const long long total = (long long)a.N * cin * a.H * a.W;
  return fd_launch(&dgrad_direct_kernel, "dgrad_direct_nhwc", dim3((unsigned)((total + 255) / 256)), dim3(256), 0, a,
                   static_cast<hipStream_t>(stream));
}
