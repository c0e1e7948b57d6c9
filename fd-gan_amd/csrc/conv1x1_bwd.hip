// conv1x1_bwd.hip -- data gradient of the dense-layer bottleneck (1x1, Cout_fwd = Cy <= 128 filters, any Cin_fwd = C <= 1024)
// fused with the backward of its BatchNorm + ReLU prologue, as a streaming kernel.
//
//   da[p][c]  = sum_k dy[p][k] * W[k][c]                  (k over the Cy forward filters)
//   v         = da * act'(bn(x[p][c]));   sums (v, v * x) per channel;   G[p][c] += gamma*rstd * v   (or T = v)
//
// The generic implicit-GEMM kernel (conv_igemm.h, MK = 1) moves 2.7-2.9 TB/s on these shapes: a 128-pixel x 128-channel
// tile per workgroup with four K steps is all prologue and epilogue, and every channel tile re-reads dy.  Here the
// work is organised around the data that is big -- x and G, 3 C values per pixel against Cy of dy:
//   * a workgroup (4 waves, 2 per CU) owns ONE 128-channel tile for its whole life -- filter fragments (32 KB) staged
//     once, the lane's 8 channels' coefficients and BatchNorm sums in registers -- and walks 64-pixel tiles of the
//     flattened N*H*W axis; the channel tiles of the same pixels run back to back on one XCD, so dy (small next to x and
//     G) comes out of that L2 for all but the first;
//   * the dy, x and G rows of a pixel tile are requested one whole tile AHEAD (two register sets used alternately), so
//     HBM reads are in flight while the previous tile's epilogue computes and stores.  (First version: pixel tile outer,
//     channel tiles inner, requests in the same iteration: every phase serialised, 570 us for a 256x256 layer while each
//     phase alone ran at HBM speed; restaging the filter per pair and the per-pair sums cost another 1.2 us of 7.8.)
//   * the epilogue is the row phase of the MK kernels: the accumulator tile goes through a wave-private LDS
//     transposition, every lane owns 8 channels of 4 pixels, whole 16-byte pieces of pixel rows in and out;
//   * BatchNorm's two sums leave as one partial row per (pixel slot, channel tile), summed in a fixed order.
// Reference: autograd of conv1 / norm1 / relu1 of torchvision's _DenseLayer as used by
// /root/reference/models/dehaze1113.py:713-724.
#include <stdlib.h>

#include <type_traits>

#include "conv_igemm.h"

namespace {

constexpr int B1_PX = 64;                        // pixels per tile (one 16-pixel MFMA column block per wave)
constexpr int B1_CT = 128;                       // channels per channel tile
constexpr int B1_TBP = B1_CT * 2 + 16;           // staging pitch of one pixel (272 B)
constexpr int B1_TB = 16 * B1_TBP;               // per-wave transposition area
// LDS-only barrier: hipcc's __syncthreads() also waits for every outstanding global load, which would drain the x / G
// rows requested ahead of the MFMAs
#define B1_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

struct Bwd1Args {
  const unsigned short* dy;   // [P][dy_pitch] bf16, Cy channels used
  int dy_pitch, Cy;
  const unsigned short* w;    // chunk32 filter image of the (flipped = same for 1x1) forward filter: A fragments
  int ntile_total;            // ceil(C / 16)
  const unsigned short* x;    // forward input (fp16), [P][x_pitch]
  int x_pitch;
  unsigned short* g;          // gradient buffer (accumulate) or dpre
  int g_pitch;
  int C, nct;                 // channels, channel tiles
  long long P;
  int ntiles;                 // P / 64
  int mode, acc;              // 1 activation only, 2 BatchNorm + activation; acc 1: G += gamma*rstd*v, 2: G = gamma*rstd*v, 0: G = v
  float slope, eps;
  const float *mean, *var, *gamma, *beta;
  float* partial;             // [pixel slots][nct * 128][2] or NULL
  int dbg;                    // FDGAN_DEBUG_PHASES (results wrong): 1 no filter loads, 2 no MFMAs, 4 no row phase, 8 no x / G loads, 16 no dy tile
};

__global__ __launch_bounds__(256, 2) void conv1x1_bwd_kernel(Bwd1Args a) {
  extern __shared__ __attribute__((aligned(16))) char b1_lds[];
  const int kch = a.Cy / 32;                                  // k chunks (<= 4)
  char* dyt = b1_lds;                                         // [64 px][Cy * 2 B], 16-byte columns XOR-swizzled by the pixel
  char* wt = dyt + B1_PX * a.Cy * 2;                          // [kch][8 tiles][1 KB] A fragments of THIS workgroup's channel tile
  char* tb0 = wt + kch * 8 * 1024;                            // 4 x transposition areas
  float* red = reinterpret_cast<float*>(tb0);                 // [4 waves][128][2]: aliases the transposition areas, used once at the end
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = lane & 15, kgl = lane >> 4;
  char* tb = tb0 + wave * B1_TB;
  // work item: (channel tile, pixel slot).  The channel tiles of one slot read the same dy pixels: XCD-aware numbering
  // keeps them on one XCD, back to back, so that dy comes out of that L2 for all but the first.
  int item = blockIdx.x;
  {
    const int per_xcd = gridDim.x >> 3;
    if (item < per_xcd * 8) item = (item & 7) * per_xcd + (item >> 3);
  }
  const int ct = item % a.nct, slot = item / a.nct, nslots = (int)gridDim.x / a.nct;
  const int c0 = ct * B1_CT;
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  const int dyrow = a.Cy * 2;                                 // bytes per staged dy pixel
  const int dcols = a.Cy / 8;                                 // 16-byte columns per dy pixel: 4, 8 or 16
  typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
  typedef __attribute__((ext_vector_type(4))) float f4_t;
  // row phase: lanes of a quad own the same 8-channel piece of 4 consecutive pixels -- the same 8 channels for the
  // whole kernel: their coefficients and BatchNorm sums live in registers
  const int piece = lane >> 2, q0 = lane & 3;
  const int cg = c0 + piece * 8;
  const bool ch_ok = cg < a.C;
  float sc8[8], sh8[8], s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = cg + e;
    sc8[e] = 1.f, sh8[e] = 0.f, s1[e] = s2[e] = 0.f;
    if (a.mode == 2 && c < a.C) {
      const float gm = a.gamma ? a.gamma[c] : 1.f, bt = a.beta ? a.beta[c] : 0.f;
      sc8[e] = gm / sqrtf(a.var[c] + a.eps);
      sh8[e] = bt - a.mean[c] * sc8[e];
    }
  }
  // ---- this channel tile's filter fragments: once
  if (!(a.dbg & 1))
    for (int f = tid; f < kch * 512; f += 256) {              // 16-byte unit of the [kch][8][64 lanes] fragment block
      const int kc = f >> 9, t8 = (f >> 6) & 7, ln = f & 63;
      const int tile16 = ct * 8 + t8;
      lds_write16(wt + f * 16, tile16 < a.ntile_total ? *reinterpret_cast<const u32x4*>(a.w + ((long long)kc * a.ntile_total + tile16) * 512 + ln * 8) : zero4);
    }
  const int my_tiles = slot < a.ntiles ? (a.ntiles - slot + nslots - 1) / nslots : 0;

  // dy rows, x / G rows of one pixel tile: requested one whole tile ahead (two register sets, used alternately)
  u32x4 dyr[4];   // one set: a tile's dy rows are written to LDS at the start of its step, the next request reuses the registers
  auto request_dy = [&](int tl) __attribute__((always_inline)) {
    const long long p0 = (long long)(slot + tl * nslots) * B1_PX;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int u = tid + i * 256;
      dyr[i] = zero4;
      if (tl < my_tiles && u < B1_PX * dcols && !(a.dbg & 16)) dyr[i] = *reinterpret_cast<const u32x4*>(a.dy + (p0 + u / dcols) * a.dy_pitch + (u % dcols) * 8);
    }
  };
  auto request = [&](int tl, u32x4 (&xv)[4], u32x4 (&gv)[4]) __attribute__((always_inline)) {
    const long long p0 = (long long)(slot + tl * nslots) * B1_PX;
    const bool live = tl < my_tiles;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long long p = p0 + wave * 16 + i * 4 + q0;
      xv[i] = gv[i] = zero4;
      if (live && ch_ok && !(a.dbg & 8)) {
        xv[i] = *reinterpret_cast<const u32x4*>(a.x + p * a.x_pitch + cg);
        if (a.acc == 1) gv[i] = *reinterpret_cast<const u32x4*>(a.g + p * a.g_pitch + cg);
      }
    }
  };
  auto step = [&](int tl, u32x4 (&xv)[4], u32x4 (&gv)[4], u32x4 (&xn)[4], u32x4 (&gn)[4]) __attribute__((always_inline)) {
    const long long p0 = (long long)(slot + tl * nslots) * B1_PX;
    B1_BARRIER();                                             // previous tile's dy fragments read by every wave
#pragma unroll
    for (int i = 0; i < 4; ++i) {   // column c16 of pixel q lands at column c16 ^ (q mod columns): conflict-free B-fragment reads
      const int u = tid + i * 256, q = u / dcols, c16 = u % dcols;
      if (u < B1_PX * dcols) lds_write16(dyt + q * dyrow + ((c16 ^ (q & (dcols - 1))) << 4), dyr[i]);
    }
    B1_BARRIER();
    request_dy(tl + 1);                                       // next tile: in flight during this tile's MFMAs and row phase
    request(tl + 1, xn, gn);
    // ---- MFMA: wave = 16 pixels x 128 channels
    f32x4 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int q = wave * 16 + m;
    if (!(a.dbg & 2))
      for (int kc = 0; kc < kch; ++kc) {
        const int c16 = kc * 4 + kgl;
        const bf16x8 bfr = __builtin_bit_cast(bf16x8, lds_read16(dyt + q * dyrow + ((c16 ^ (q & (dcols - 1))) << 4)));
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const bf16x8 afr = __builtin_bit_cast(bf16x8, lds_read16(wt + ((kc * 8 + j) * 64 + lane) * 16));
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr, bfr, acc[j], 0, 0, 0);
        }
      }
    // ---- row phase: transpose through the wave's staging area, mask, sums, G (+)= gamma*rstd*v (or store v)
    if (a.dbg & 4) return;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const u32x2 bits = __builtin_bit_cast(u32x2, __builtin_convertvector((f4_t){acc[j][0], acc[j][1], acc[j][2], acc[j][3]}, bf16x4_t));
      *reinterpret_cast<u32x2*>(tb + m * B1_TBP + j * 32 + kgl * 8) = bits;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ql = i * 4 + q0;
      const f32x8 da = __builtin_convertvector(__builtin_bit_cast(bf16x8, lds_read16(tb + ql * B1_TBP + piece * 16)), f32x8);
      const f32x8 fx = fd_cvt8<FmtA>(xv[i]);                  // the forward input: fp16
      f32x8 o = __builtin_convertvector(__builtin_bit_cast(bf16x8, gv[i]), f32x8);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float pre = fmaf(fx[e], sc8[e], sh8[e]);
        const float v = (cg + e < a.C) ? da[e] * (pre > 0.f ? 1.f : a.slope) : 0.f;
        s1[e] += v;
        s2[e] += v * fx[e];
        o[e] = a.acc ? fmaf(sc8[e], v, o[e]) : v;
      }
      if (ch_ok) *reinterpret_cast<u32x4*>(a.g + (p0 + wave * 16 + ql) * a.g_pitch + cg) = __builtin_bit_cast(u32x4, __builtin_convertvector(o, bf16x8));
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };

  u32x4 xa[4], ga[4], xb[4], gb[4];
  request_dy(0);
  request(0, xa, ga);
  for (int tl = 0; tl < my_tiles; tl += 2) {
    step(tl, xa, ga, xb, gb);
    if (tl + 1 < my_tiles) step(tl + 1, xb, gb, xa, ga);
  }
  if (a.partial != nullptr) {   // quad -> lane 0 of the quad, then the four waves in a fixed order
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      s1[e] += __shfl_xor(s1[e], 1, 64);
      s2[e] += __shfl_xor(s2[e], 1, 64);
      s1[e] += __shfl_xor(s1[e], 2, 64);
      s2[e] += __shfl_xor(s2[e], 2, 64);
    }
    B1_BARRIER();                                             // every wave is through its last row phase (red aliases tb)
    if (q0 == 0)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        red[(wave * 128 + piece * 8 + e) * 2] = s1[e];
        red[(wave * 128 + piece * 8 + e) * 2 + 1] = s2[e];
      }
    B1_BARRIER();
    const int c = tid >> 1, which = tid & 1;
    const float t = (red[(0 * 128 + c) * 2 + which] + red[(1 * 128 + c) * 2 + which]) +
                    (red[(2 * 128 + c) * 2 + which] + red[(3 * 128 + c) * 2 + which]);
    a.partial[((long long)slot * a.nct * B1_CT + c0 + c) * 2 + which] = t;
  }
}

// ---- the FUSED form (Cy == 128): data gradient AND weight gradient of the bottleneck in one pass.
// The weight gradient dW[co][ci] = sum_p dy[p][co] * act(bn(x))[p][ci] needs exactly what the data-gradient kernel has on chip
// for a pixel tile -- the dy tile in LDS and act(bn(x)) in the row phase's registers -- so the workgroup that owns a
// 128-channel tile also accumulates its [128 co][128 ci] slice of dW over all its pixel tiles, and conv_wgrad1x1_tr's separate
// pass over dy and x (P (C + 128) 2 bytes of reads, a quarter of the pair's HBM traffic) disappears.
// Workgroup = 8 waves, one per CU (the 64 extra accumulator registers of a 4-wave workgroup did not fit beside the two x / G
// register sets): wave (w & 3, w >> 2) = (16-pixel quarter, 64-channel half).
//   data gradient     16 pixels x 64 channels per wave (4 accumulator tiles); row phase: a lane owns 8 channels of 2 pixels
//   weight gradient   wave (w & 1, w >> 1) owns 4 cout tiles x 2 cin tiles (8 accumulator tiles), operands by transpose reads
//                     from the dy tile and the activated tile (both [pixel][256 B], conv_wgrad1x1_tr's swizzle: conflict-free
//                     for those reads and for the data gradient's 16-byte B fragments)
// ONE barrier per 64-pixel step: the dy tile is triple-, the activated tile double-buffered, and the weight gradient of tile
// t - 1 is issued in step t, right before the row phase -- its 16 MFMAs run in the matrix pipe while the row phase's VALU work
// issues (first version: single buffers, three barriers, weight gradient after the row phase: 134 us per launch for work the
// data-gradient kernel alone did in 104).
constexpr int F1_TBP = 64 * 2 + 16;              // transposition pitch of one pixel (64 channels)
constexpr int F1_TB = 16 * F1_TBP;               // per wave
constexpr int F1_TILE = B1_PX * 256;             // a [64 px][128 ch] bf16 tile
constexpr int F1_LDS = 3 * F1_TILE + 4 * 8 * 1024 + 8 * F1_TB + 2 * F1_TILE + 1024 + 1024;   // dy tiles, filter, transposition, activated tiles, dy_affine coefficients, BatchNorm scale / shift

typedef short f1_s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8 f1_trfrag(const char* p0, const char* p1) {
  const f1_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) f1_s16x4*)(p0));
  const f1_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) f1_s16x4*)(p1));
  return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}
__device__ __forceinline__ int f1_off(int pix, int c16g) {   // 32-byte (16-channel) group c16g of pixel pix in a [pixel][256 B] tile
  return pix * 256 + ((c16g ^ ((pix & 3) | (((pix >> 3) & 1) << 2))) << 5);
}

struct Bwdw1Args {
  Bwd1Args b;
  float* wpart;               // [pixel slots][128 filters][C] partial weight gradients
  // pending linear part of the BatchNorm backward of dy's OWN producer (the growth conv's norm2): dy' = dy + cB * yb + cC per
  // channel, yb = that norm's input (the bottleneck activation).  Applied while the dy tile is staged, so the separate
  // read-read-write pass over the 128-channel gradient buffer (affine_accumulate) is not needed.  NULL: dy is final.
  const unsigned short* yb;   // [P][yb_pitch], 128 channels, fp16
  int yb_pitch;
  const float *cB, *cC;       // [128]
};

// AFFINE: dy_affine given; ACC: 0 G = v, 1 G += sc * v (G is read), 2 G = sc * v.  Template parameters, and no branch around a
// global load in the steady-state loop (clamped addresses and a peeled last tile instead): with the uniform `if`s the first
// version had there, hipcc's s_waitcnt vmcnt counts degraded to vmcnt(0) -- the dy rows of the NEXT tile could not be waited
// for without also waiting for its x and G rows, requested in the same breath for the END of that step, so every step
// began by draining the whole prefetch.
template <bool AFFINE, int ACC>
__global__ __launch_bounds__(512) void conv1x1_bwdw_kernel(Bwdw1Args aa) {
  const Bwd1Args& a = aa.b;
  extern __shared__ __attribute__((aligned(16))) char b1_lds[];
  char* dyt0 = b1_lds;                                        // 3 x [64 px][256 B]: tile t lives in t mod 3
  char* wt = dyt0 + 3 * F1_TILE;                              // [4 k chunks][8 tiles][1 KB] A fragments of this channel tile
  char* tb0 = wt + 4 * 8 * 1024;                              // 8 x transposition areas
  char* at0 = tb0 + 8 * F1_TB;                                // 2 x activated tile [64 px][256 B]: tile t lives in t mod 2
  float* cf = reinterpret_cast<float*>(at0 + 2 * F1_TILE);    // dy_affine: cB[128], cC[128] (read per step: 16 registers less)
  float* scf = cf + 256;                                      // this channel tile's BatchNorm scale[128], shift[128]: read per row phase
                                                              // (16 registers less: the third x / G set of the two-tile prefetch needs them)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, kgl = lane >> 4;
  const int pq = wave & 3, chh = wave >> 2;                   // 16-pixel quarter, 64-channel half
  char* tb = tb0 + wave * F1_TB;
  int item = blockIdx.x;
  {
    const int per_xcd = gridDim.x >> 3;
    if (item < per_xcd * 8) item = (item & 7) * per_xcd + (item >> 3);
  }
  const int ct = item % a.nct, slot = item / a.nct, nslots = (int)gridDim.x / a.nct;
  const int c0 = ct * B1_CT;
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
  typedef __attribute__((ext_vector_type(4))) float f4_t;
  // row phase: lane -> 8-channel piece lane & 7 of the wave's 64 channels, pixels lane >> 3 and 8 + (lane >> 3) of its 16
  const int piece = lane & 7, ql0 = lane >> 3;
  const int cg = c0 + chh * 64 + piece * 8;
  const bool ch_ok = cg < a.C;
  const int cg_ld = ch_ok ? cg : 0;                           // lanes past the last channel work on channels 0-7 (their G store is skipped,
                                                              // their sums and weight-gradient columns are never read; C % 8 == 0).
  // (Round 4: neither letting them read their own address up to the end of the last 128-byte line, nor parking them all on ONE
  // address per wave instruction, changed anything -- C = 96 takes 270-275 us at 256^2 whatever they fetch, the time of C = 128
  // (270 us): a channel tile costs its step count, not its bytes; tools/bwdw_one.py.)
  f32x2 s1[4], s2[4];                                         // BatchNorm's two sums of the lane's 8 channels, as 4 pairs (fd_row8)
#pragma unroll
  for (int k = 0; k < 4; ++k) s1[k] = s2[k] = f32x2{0.f, 0.f};
  if (tid < 128) {                                            // scale / shift of the tile's 128 channels -> LDS
    const int c = c0 + tid < a.C ? c0 + tid : 0;
    float sc = 1.f, sh = 0.f;
    if (a.mode == 2) {
      const float gm = a.gamma ? a.gamma[c] : 1.f, bt = a.beta ? a.beta[c] : 0.f;
      sc = gm / sqrtf(a.var[c] + a.eps);
      sh = bt - a.mean[c] * sc;
    }
    scf[tid] = sc, scf[128 + tid] = sh;
  }
  const float* scl = scf + chh * 64 + piece * 8;              // the lane's 8 channels
  for (int f = tid; f < 4 * 512; f += 512) {                  // this channel tile's filter fragments: once
    const int kc = f >> 9, t8 = (f >> 6) & 7, ln = f & 63;
    const int tile16 = ct * 8 + t8;
    lds_write16(wt + f * 16, tile16 < a.ntile_total ? *reinterpret_cast<const u32x4*>(a.w + ((long long)kc * a.ntile_total + tile16) * 512 + ln * 8) : zero4);
  }
  const int my_tiles = slot < a.ntiles ? (a.ntiles - slot + nslots - 1) / nslots : 0;
  // weight-gradient side: lane (g, i) of a transpose read: k rows (pixels) 8 g + (i >> 2) (+ 4), 4-channel piece i & 3
  const int tg = lane >> 4, ti = lane & 15;
  const int kpix = 8 * tg + (ti >> 2), piece8 = (ti & 3) * 8;
  const int wco = (wave & 1) * 4, wci = (wave >> 1) * 2;      // first cout / cin 16-channel tile of this wave
  f32x4 wacc[4][2];
#pragma unroll
  for (int c = 0; c < 4; ++c) wacc[c][0] = wacc[c][1] = f32x4{0.f, 0.f, 0.f, 0.f};

  // dy rows (2 units per thread), x / G rows (2 units) of a pixel tile: requested one whole tile ahead, two register sets
  u32x4 dyr[2], ybr[2];
  if (AFFINE && tid < 256) cf[tid] = tid < 128 ? aa.cB[tid] : aa.cC[tid - 128];
  __syncthreads();                                            // cf (read BEFORE the first step's barrier) and scf
  const float* cfl = cf + (tid & 15) * 8;                     // the thread's dy column is the same for both units
  // addresses: a wave-uniform tile base (scalar arithmetic) + the lane's 32-bit element offset inside the 64-pixel tile, fixed
  // for the whole kernel (the first version multiplied 64-bit pixel indices per load: v_mad_u64_u32 / v_lshl_add_u64 per access)
  unsigned dyo[2], ybo[2], xo[2], go[2], ato[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int u = tid + i * 512;                              // dy / yb unit: (pixel u / 16, 16-byte column u % 16)
    dyo[i] = (unsigned)((u >> 4) * a.dy_pitch + (u & 15) * 8);
    ybo[i] = AFFINE ? (unsigned)((u >> 4) * aa.yb_pitch + (u & 15) * 8) : 0u;
    const int pl = pq * 16 + i * 8 + ql0;                     // row-phase unit: pixel pl of the tile, this lane's 8 channels
    xo[i] = (unsigned)(pl * a.x_pitch + cg_ld);
    go[i] = (unsigned)(pl * a.g_pitch + cg_ld);
    const int c16 = chh * 8 + piece;                          // 16-byte column of the 128-channel activated tile
    ato[i] = (unsigned)(f1_off(pl, c16 >> 1) + ((c16 & 1) << 4));
  }
  auto request_dy = [&](int tl) __attribute__((always_inline)) {   // tl < my_tiles
    const long long p0 = (long long)(slot + tl * nslots) * B1_PX;
    const unsigned short* dyt_g = a.dy + p0 * a.dy_pitch;
    const unsigned short* ybt_g = AFFINE ? aa.yb + p0 * aa.yb_pitch : nullptr;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      dyr[i] = *reinterpret_cast<const u32x4*>(dyt_g + dyo[i]);
      if (AFFINE) ybr[i] = *reinterpret_cast<const u32x4*>(ybt_g + ybo[i]);
    }
  };
  auto request = [&](int tl, u32x4 (&xv)[2], u32x4 (&gv)[2]) __attribute__((always_inline)) {   // tl < my_tiles
    const long long p0 = (long long)(slot + tl * nslots) * B1_PX;
    const unsigned short* xt_g = a.x + p0 * a.x_pitch;
    const unsigned short* gt_g = a.g + p0 * a.g_pitch;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      xv[i] = *reinterpret_cast<const u32x4*>(xt_g + xo[i]);
      gv[i] = zero4;
      if (ACC == 1) gv[i] = *reinterpret_cast<const u32x4*>(gt_g + go[i]);
    }
  };
  // weight gradient of one tile: dW[128 co][128 ci tile] += dy^T (64 px x 128 co) * act (64 px x 128 ci)
  auto wgrad_tile = [&](const char* dyt, const char* at) __attribute__((always_inline)) {
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      const int pa = 32 * sub + kpix;
      bf16x8 af[4], bf[2];
#pragma unroll
      for (int c = 0; c < 4; ++c) af[c] = f1_trfrag(dyt + f1_off(pa, wco + c) + piece8, dyt + f1_off(pa + 4, wco + c) + piece8);
#pragma unroll
      for (int j = 0; j < 2; ++j) bf[j] = f1_trfrag(at + f1_off(pa, wci + j) + piece8, at + f1_off(pa + 4, wci + j) + piece8);
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int j = 0; j < 2; ++j) wacc[c][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[c], bf[j], wacc[c][j], 0, 0, 0);
    }
  };
  unsigned dyl[2];                                             // LDS offset of the thread's two dy units inside a staged tile
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int u = tid + i * 512, q = u >> 4, c16 = u & 15;
    dyl[i] = (unsigned)(f1_off(q, c16 >> 1) + ((c16 & 1) << 4));
  }
  // One step = one 64-pixel tile; the next tile's rows are requested right after the step's barrier.  (Requesting x / G TWO tiles
  // ahead in a third register set was tried in round 3: 76-368 B of scratch per lane whatever else was moved to LDS, and a
  // scratch reload is a vmcnt(0): 2.4-3.6 TB/s instead of 4.1-5.6.  Warming L2 two tiles ahead instead -- one dword per 128-byte
  // line of the x / G rows, one register -- changed nothing, 4.60 against 4.72 TB/s over the 42 shapes: the kernel is not short of
  // bytes in flight.  What the per-shape table does show: prefixes of C = 64 k channels stream at 4.9-6.0 TB/s, C = 64 k + 32 at
  // 4.0-4.5 -- rows that end in half a 128-byte line -- against 6.1 TB/s for torch's own 2-read-1-write add on this board.)
  int d3 = 0;                                                  // tl mod 3
  auto step = [&](int tl, u32x4 (&xv)[2], u32x4 (&gv)[2], u32x4 (&xn)[2], u32x4 (&gn)[2], auto last) __attribute__((always_inline)) {
    const long long p0 = (long long)(slot + tl * nslots) * B1_PX;
    char* dyt = dyt0 + d3 * F1_TILE;
    char* at = at0 + (tl & 1) * F1_TILE;
    const char* dyp = dyt0 + (d3 == 0 ? 2 : d3 - 1) * F1_TILE;   // the previous tile's
    const char* atp = at0 + ((tl & 1) ^ 1) * F1_TILE;
    d3 = d3 == 2 ? 0 : d3 + 1;
    // no barrier here: slot t mod 3 was last read by the weight gradient of tile t - 3, issued in step t - 2 -- every wave
    // is past that once it has passed the barrier of step t - 1
    f32x8 cb8, cc8;
    if (AFFINE) {
      cb8 = *reinterpret_cast<const f32x8*>(cfl);
      cc8 = *reinterpret_cast<const f32x8*>(cfl + 128);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      u32x4 v = dyr[i];
      if (AFFINE) {     // dy' = dy + cB * yb + cC, rounded to bf16 exactly as the separate pass stored it
        f32x8 f = __builtin_convertvector(__builtin_bit_cast(bf16x8, v), f32x8);
        const f32x8 y8 = fd_cvt8<FmtA>(ybr[i]);               // the bottleneck activation: fp16
        f = __builtin_elementwise_fma(cb8, y8, f + cc8);        // packed fp32: 8 instructions
        v = fd_pk8_sr(f, (unsigned)p0 * 16u + (unsigned)(tid + i * 512));   // dy is on the bf16 grid already: stochastic rounding (common.h)
      }
      lds_write16(dyt + dyl[i], v);
    }
    B1_BARRIER();                                             // this tile's dy rows and the previous tile's activated rows are in place
    if (!decltype(last)::value) {
      request_dy(tl + 1);                                     // next tile: in flight during this tile's MFMAs and row phase
      request(tl + 1, xn, gn);
    }
    // ---- data gradient MFMA: wave = 16 pixels x 64 channels
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int q = pq * 16 + m;
    // raised priority from the data-gradient MFMAs to the weight-gradient block: the co-resident wave's loads and row-phase VALU
    // yield to the MFMA issue (round 4, tools/bwdw_one.py over the 42 shapes: 4.92 / 4.99 ms without, 4.85 / 4.86 ms with)
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
      const int c16 = kc * 4 + kgl;
      const bf16x8 bfr = __builtin_bit_cast(bf16x8, lds_read16(dyt + f1_off(q, c16 >> 1) + ((c16 & 1) << 4)));
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bf16x8 afr = __builtin_bit_cast(bf16x8, lds_read16(wt + ((kc * 8 + chh * 4 + j) * 64 + lane) * 16));
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr, bfr, acc[j], 0, 0, 0);
      }
    }
    // ---- row phase: transpose through the wave's staging area, mask, sums, G (+)= gamma*rstd*v (or store v); keep act(bn(x))
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const u32x2 bits = __builtin_bit_cast(u32x2, __builtin_convertvector((f4_t){acc[j][0], acc[j][1], acc[j][2], acc[j][3]}, bf16x4_t));
      *reinterpret_cast<u32x2*>(tb + m * F1_TBP + j * 32 + kgl * 8) = bits;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (tl > 0 && !(a.dbg & 32)) wgrad_tile(dyp, atp);        // its MFMAs execute while the row phase below issues
    __builtin_amdgcn_s_setprio(0);
    unsigned short* gst = a.g + p0 * a.g_pitch;                 // wave-uniform
    f32x2 sc2[4], sh2[4];
    {
      const f32x4 sa = *reinterpret_cast<const f32x4*>(scl), sb = *reinterpret_cast<const f32x4*>(scl + 4);
      const f32x4 ha = *reinterpret_cast<const f32x4*>(scl + 128), hb = *reinterpret_cast<const f32x4*>(scl + 132);
      sc2[0] = f32x2{sa[0], sa[1]}, sc2[1] = f32x2{sa[2], sa[3]}, sc2[2] = f32x2{sb[0], sb[1]}, sc2[3] = f32x2{sb[2], sb[3]};
      sh2[0] = f32x2{ha[0], ha[1]}, sh2[1] = f32x2{ha[2], ha[3]}, sh2[2] = f32x2{hb[0], hb[1]}, sh2[3] = f32x2{hb[2], hb[3]};
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int ql = i * 8 + ql0;
      const f32x8 da = __builtin_convertvector(__builtin_bit_cast(bf16x8, lds_read16(tb + ql * F1_TBP + piece * 16)), f32x8);
      const f32x8 fx = fd_cvt8<FmtA>(xv[i]);                  // the forward input: fp16
      f32x8 o = __builtin_convertvector(__builtin_bit_cast(bf16x8, gv[i]), f32x8);
      f32x8 act;
      fd_row8<ACC, true>(da, fx, o, sc2, sh2, 1.f, a.slope, s1, s2, act);      // act: what the forward conv saw
      // (lanes past the last channel computed on channels 0-7: finite values in weight-gradient columns that are never stored)
      if (!(a.dbg & 64)) lds_write16(at + ato[i], __builtin_bit_cast(u32x4, __builtin_convertvector(act, bf16x8)));
      if (ch_ok) *reinterpret_cast<u32x4*>(gst + go[i]) = __builtin_bit_cast(u32x4, __builtin_convertvector(o, bf16x8));
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };

  u32x4 xa[2], ga[2], xb[2], gb[2];
  const std::integral_constant<bool, false> MORE;
  const std::integral_constant<bool, true> LAST;
  request_dy(0);                                              // my_tiles >= 1: there are never more pixel slots than tiles
  request(0, xa, ga);
  int tl = 0;
  for (; tl + 2 < my_tiles; tl += 2) {                        // steady state: both steps have a next tile to request
    step(tl, xa, ga, xb, gb, MORE);
    step(tl + 1, xb, gb, xa, ga, MORE);
  }
  if (tl + 2 == my_tiles) {
    step(tl, xa, ga, xb, gb, MORE);
    step(tl + 1, xb, gb, xa, ga, LAST);
  } else {
    step(tl, xa, ga, xb, gb, LAST);
  }
  B1_BARRIER();                                               // the last tile's activated rows are in place
  if (my_tiles > 0) wgrad_tile(dyt0 + ((my_tiles - 1) % 3) * F1_TILE, at0 + ((my_tiles - 1) & 1) * F1_TILE);
  B1_BARRIER();                                               // every wave is through its last fragment reads
  if (a.partial != nullptr) {   // lanes 8 apart own the same channels; then the four pixel quarters in a fixed order
    float t1[8], t2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      t1[e] = s1[e >> 1][e & 1], t2[e] = s2[e >> 1][e & 1];
#pragma unroll
      for (int d = 8; d < 64; d <<= 1) {
        t1[e] += __shfl_xor(t1[e], d, 64);
        t2[e] += __shfl_xor(t2[e], d, 64);
      }
    }
    float* red = reinterpret_cast<float*>(tb0);               // [8 waves][64][2]
    if (ql0 == 0)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        red[(wave * 64 + piece * 8 + e) * 2] = t1[e];
        red[(wave * 64 + piece * 8 + e) * 2 + 1] = t2[e];
      }
    B1_BARRIER();
    if (tid < 256) {
      const int c = tid >> 1, which = tid & 1, h = c >> 6, cl = c & 63;   // channel half h: waves 4 h .. 4 h + 3
      const float t = (red[((4 * h + 0) * 64 + cl) * 2 + which] + red[((4 * h + 1) * 64 + cl) * 2 + which]) +
                      (red[((4 * h + 2) * 64 + cl) * 2 + which] + red[((4 * h + 3) * 64 + cl) * 2 + which]);
      a.partial[((long long)slot * a.nct * B1_CT + c0 + c) * 2 + which] = t;
    }
    B1_BARRIER();
  }
  // ---- weight-gradient partial: D layout column (lane & 15) = cin, rows (lane >> 4) * 4 + r = cout -> LDS [128][128] fp32 -> rows
  float* out = reinterpret_cast<float*>(b1_lds);              // 64 KB: dy tile, filter image and transposition areas are dead
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) out[((wco + c) * 16 + tg * 4 + r) * 128 + (wci + j) * 16 + ti] = wacc[c][j][r];
  __syncthreads();
  float* dwp = aa.wpart + (long long)slot * 128 * a.C;
  for (int u = tid; u < 128 * 32; u += 512) {                 // 16-byte pieces of the [128][128] tile
    const int co = u >> 5, c4 = (u & 31) * 4;
    const f32x4 v = *reinterpret_cast<const f32x4*>(out + co * 128 + c4);
    float* d = dwp + (long long)co * a.C + c0 + c4;
    if (c0 + c4 + 4 <= a.C && (a.C & 3) == 0) {
      *reinterpret_cast<f32x4*>(d) = v;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (c0 + c4 + e < a.C) d[e] = v[e];
    }
  }
}

}  // namespace

bool conv1x1_bwd_fits(const FdTensor* dy, const FdTensor* fwd_x, const FdTensor* dpre) {
  auto dense = [](const FdTensor* t) {
    return t->stride[3] == 1 && t->stride[1] == t->w * t->stride[2] && t->stride[0] == t->h * t->stride[1] && t->stride[2] % 8 == 0 &&
           ((uintptr_t)t->ptr & 15) == 0;
  };
  const long long P = dpre->n * dpre->h * dpre->w;
  return (dy->c == 32 || dy->c == 64 || dy->c == 128) && dense(dy) && dense(fwd_x) && dense(dpre) && P % B1_PX == 0 && dpre->c % 8 == 0 &&
         dpre->c <= 1024 && dpre->stride[2] >= (dpre->c + 7) / 8 * 8 && fwd_x->stride[2] >= (dpre->c + 7) / 8 * 8 &&
         P * fwd_x->stride[2] < (1ll << 40) && FD_TUNE_GETENV("FDGAN_DEBUG_NO_BWD1X1S") == nullptr;
}

/* rows_out / cpad_out: shape of the partial-sum block written when `partial` is given. */
int conv1x1_bwd_launch(const FdTensor* dy, const void* w_packed, const FdTensor* fwd_x, const FdPrologue* pro, const FdTensor* dpre,
                       int accumulate, float* partial, long long capacity_floats, long long* rows_out, long long* cpad_out,
                       hipStream_t stream, float* wpart, long long wpart_floats, long long* wsplit_out, const FdTensor* dy_affine_x,
                       const float* dy_affine_b, const float* dy_affine_c) {
  Bwd1Args a{};
  a.dy = static_cast<const unsigned short*>(dy->ptr), a.dy_pitch = (int)dy->stride[2], a.Cy = (int)dy->c;
  a.w = static_cast<const unsigned short*>(w_packed);
  a.C = (int)dpre->c, a.ntile_total = (a.C + 15) / 16, a.nct = (a.C + B1_CT - 1) / B1_CT;
  a.x = static_cast<const unsigned short*>(fwd_x->ptr), a.x_pitch = (int)fwd_x->stride[2];
  a.g = static_cast<unsigned short*>(dpre->ptr), a.g_pitch = (int)dpre->stride[2];
  a.P = dpre->n * dpre->h * dpre->w, a.ntiles = (int)(a.P / B1_PX);
  const bool norm = pro && pro->mean;
  const int act = pro ? pro->act : FD_ACT_NONE;
  a.mode = norm ? 2 : 1, a.acc = accumulate;
  a.slope = act == FD_ACT_RELU ? 0.f : (act == FD_ACT_LEAKY02 ? 0.2f : 1.f);
  if (norm) a.mean = pro->mean, a.var = pro->var, a.gamma = pro->gamma, a.beta = pro->beta, a.eps = pro->eps;
  const bool fused = wpart != nullptr;                         // one 8-wave workgroup per CU instead of two 4-wave ones
  if (fused && a.Cy != 128) return 1;
  long long nslots = (fused ? fd_cus(256) : 2 * fd_cus(256)) / a.nct;   // one 8-wave / two 4-wave workgroups per CU of the budget
  if (nslots < 1) nslots = 1;
  if (nslots > a.ntiles) nslots = a.ntiles;
  const long long grid = nslots * a.nct;
  a.partial = norm ? partial : nullptr;
  if (norm) FD_REQUIRE(partial && nslots * a.nct * 256 <= capacity_floats, "conv2d_bwd_data: workspace too small (%lld floats needed)", nslots * a.nct * 256);
  static const char* ph = FD_TUNE_GETENV("FDGAN_DEBUG_PHASES");
  a.dbg = ph ? atoi(ph) : 0;
  if (rows_out) *rows_out = nslots;
  if (cpad_out) *cpad_out = a.nct * 128;
  const unsigned lds = (unsigned)(B1_PX * a.Cy * 2 + (a.Cy / 32) * 8 * 1024 + 4 * B1_TB);
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) FD_FAIL(FD_ELAUNCH, "hipFuncSetAttribute(conv1x1_bwd): %s", hipGetErrorString(e));
    attr_done = true;
  }
  if (fused) {   // 1: the caller runs the two separate kernels instead
    if (nslots * 128 * a.C > wpart_floats) return 1;
    if (wsplit_out) *wsplit_out = nslots;
    Bwdw1Args aa{a, wpart, nullptr, 0, nullptr, nullptr};
    if (dy_affine_x != nullptr) {
      aa.yb = static_cast<const unsigned short*>(dy_affine_x->ptr), aa.yb_pitch = (int)dy_affine_x->stride[2];
      aa.cB = dy_affine_b, aa.cC = dy_affine_c;
    }
    void (*kern)(Bwdw1Args) = nullptr;
    const bool aff = aa.yb != nullptr;
    switch (a.acc) {
      case 0: kern = aff ? &conv1x1_bwdw_kernel<true, 0> : &conv1x1_bwdw_kernel<false, 0>; break;
      case 1: kern = aff ? &conv1x1_bwdw_kernel<true, 1> : &conv1x1_bwdw_kernel<false, 1>; break;
      default: kern = aff ? &conv1x1_bwdw_kernel<true, 2> : &conv1x1_bwdw_kernel<false, 2>; break;
    }
    static bool attr_f[6] = {false, false, false, false, false, false};
    const int ki = (a.acc == 0 ? 0 : (a.acc == 1 ? 1 : 2)) * 2 + (aff ? 1 : 0);
    if (!attr_f[ki]) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) FD_FAIL(FD_ELAUNCH, "hipFuncSetAttribute(conv1x1_bwdw): %s", hipGetErrorString(e));
      attr_f[ki] = true;
    }
    return fd_launch(kern, "conv1x1_bwd_wgrad_stream", dim3((unsigned)grid), dim3(512), F1_LDS, aa, stream);
  }
  return fd_launch(&conv1x1_bwd_kernel, "conv1x1_bwd_stream", dim3((unsigned)grid), dim3(256), lds, a, stream);
}
