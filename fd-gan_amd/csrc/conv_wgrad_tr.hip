// conv_wgrad_tr.hip -- weight gradient of stride-1 k x k convs with gfx950's LDS transpose read: the dense-layer
// growth conv (3x3, pad 1, 32 filters, Cin a multiple of 128) and the Fusion-discriminator's 4x4 144 -> 288 conv.
//
//   dW[co][ci][ky][kx] = sum over (n, y, x) of dy[n][y][x][co] * a[n][y + ky - 1][x + kx - 1][ci],
//   a = relu(bn(x)) recomputed from the raw fp16 input as the forward conv staged it (rounded to bf16 here: dy is bf16; zero padding)
//
// Both MFMA operands want "8 consecutive k (= pixels) of one channel" per lane, while memory is pixel-major
// with channels contiguous.  The first kernels (conv_bwd.hip) transposed in registers (v_perm_b32 +
// ds_write_b64) and kept three kx-shifted copies of every input row so that the shifted B fragments stayed
// 16-byte aligned; they spent their time in that staging and in exposed load latency (650 us per 256x256
// layer, 10x the HBM time).  ds_read_b64_tr_b16 takes a [4 k][16 channel] block in its natural layout -- each
// of 16 lanes supplies the address of one 8-byte piece, lane i receives column i -- so:
//   * LDS holds input rows as they are in memory, [pixel][channel], written once with plain ds_write_b128;
//   * a tap is a pixel offset in the read address: no copies, all nine taps read the same rows;
//   * per output row a workgroup stages ONE new input row (66 pixels x 128 channels) and one dy row, both
//     loaded into registers one step ahead (in flight during the MFMAs), one barrier per row.
// Workgroup = 8 waves: (image, 64-pixel column block, row segment, 128-channel slice of Cin).  Wave w owns
// cin tile w (16 channels) x both cout tiles x 9 taps = 18 accumulator tiles; per 32-pixel k-sub it reads
// 2 A fragments (dy) + 9 B fragments (one per tap) for 18 MFMAs.
// Bank conflicts: the 32-byte channel group of a pixel is XORed with (pix & 3) | ((pix >> 3) & 1) << 2 (x rows,
// 256 B per pixel) / ((pix >> 3) & 1) (dy rows, 64 B per pixel), which makes the eight pixels a half-wave
// touches (p..p+3, p+8..p+11, any tap shift) fall in eight different 32-byte bank groups.  With 9 cin tiles (288 B
// per pixel) consecutive pixels already rotate through the groups; every second block of 8 pixels is shifted 128 B.
// Reference: autograd of the growth conv of torchvision's _DenseLayer as used by
// /root/reference/models/dehaze1113.py:713-724 (dense_block1-3).
#include <stdlib.h>

#include "conv_igemm.h"

namespace {

constexpr int G3_PB = 64;                       // output pixels per row step

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

// KS x KS stride-1 conv; a workgroup owns KYN filter rows (KYN * KS taps), NW 16-channel cin tiles (one per wave) and
// 2 cout tiles (32 filters): 2 * KYN * KS accumulator tiles per wave
template <int KS, int KYN, int NW, int NCO = 2>
struct G3Cfg {
  static constexpr int KK = KS * KS;
  static constexpr int TAPS = KYN * KS;
  static_assert(KS % KYN == 0, "filter rows split evenly over workgroups");
  static constexpr int NT = 64 * NW;
  static constexpr int XPIX = G3_PB + KS - 1;            // staged input pixels per row
  static constexpr int XPB = NW * 32;                    // bytes per staged pixel
  // bank spreading (see the header comment): power-of-two pixel pitch -> XOR of the 32-byte group; odd number of
  // groups -> every second block of 8 pixels shifted by 128 bytes
  static constexpr bool XOR_SWZ = (NW & (NW - 1)) == 0;
  static_assert(XOR_SWZ ? NW == 8 : (NW & 1) == 1, "bank-conflict-free layouts exist for 8 or an odd number of cin tiles");
  static constexpr int XROW_B = XPIX * XPB + (XOR_SWZ ? 0 : (XPIX / 8 + 1) * 128);
  static constexpr int XSLOTS = KYN + 1;                 // KYN rows in use + the row being written
  static constexpr int DROW_B = G3_PB * NCO * 32;        // NCO x 16 filters x 2 B per pixel
  static constexpr int DPP = NCO * 2;                    // 16-byte pieces per dy pixel
  static constexpr int SC_OFF = XSLOTS * XROW_B + 2 * DROW_B;
  static constexpr int LDS = SC_OFF + 2 * NW * 16 * 4;
  static constexpr int XUNITS = XPIX * NW * 2;           // 16-byte units per staged row
  static constexpr int XK = (XUNITS + NT - 1) / NT;      // units per thread
  __device__ static __forceinline__ int xoff(int pix, int c16) {
    if constexpr (XOR_SWZ) return pix * XPB + ((c16 ^ ((pix & 3) | (((pix >> 3) & 1) << 2))) << 5);
    return pix * XPB + (pix >> 3) * 128 + (c16 << 5);
  }
};

__device__ __forceinline__ bf16x8 g3_frag(const char* p0, const char* p1) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p1));
  return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}
__device__ __forceinline__ int g3_doff(int pix, int c16) {
  return pix * 64 + ((c16 ^ ((pix >> 3) & 1)) << 5);
}
// dy rows with NCO cout tiles: [pixel][NCO x 32 B].  NCO = 4 (round 5): a transpose read touches pixels {8 g + j} and {8 g + 4 + j}
// (g < 4, j < 4) of ONE 32-byte group; at a 128-byte pitch pixels of equal parity share half a bank row, so the group is rotated by
// ((pix >> 1) & 1) | ((pix >> 3) & 1) << 1: the four equal-parity pixels of a half-wave land in four different 32-byte bank groups.
template <int NCO>
__device__ __forceinline__ int g3_doffn(int pix, int c16) {
  if constexpr (NCO == 2) return g3_doff(pix, c16);
  else return pix * 128 + ((c16 ^ (((pix >> 1) & 1) | (((pix >> 3) & 1) << 1))) << 5);
}

// NCO: cout tiles per workgroup.  2: the original split (32 filters, grid z = filter groups x filter-row groups).  4 (round 5, 3x3 only):
// 64 filters per workgroup -- 13 fragment reads per 72 MFMAs instead of 11 per 36, and every input row is re-staged by half as many
// filter groups (D's 72 -> 144: 3 instead of 5) -- whose partial sums are written as TWO 32-filter groups of the original layout, so
// the reduction kernels do not know the difference.
template <int KS, int KYN, int NW, int NCO = 2>
__global__ __launch_bounds__(64 * NW) void conv_wgrad_tr_kernel(WgradRowsArgs a) {
  using C = G3Cfg<KS, KYN, NW, NCO>;
  static_assert(NCO == 2 || (NCO == 4 && KS == KYN), "64-filter workgroups: all filter rows in one workgroup");
  extern __shared__ __attribute__((aligned(16))) char g3_lds[];
  char* Xs = g3_lds;
  char* Ds = g3_lds + C::XSLOTS * C::XROW_B;
  float* sc_s = reinterpret_cast<float*>(g3_lds + C::SC_OFF);
  float* sh_s = sc_s + NW * 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int KYG = KS / KYN;   // filter-row groups
  const int ci0 = blockIdx.x * (NW * 16), co0 = ((int)blockIdx.z / KYG) * (NCO * 16), ky0 = ((int)blockIdx.z % KYG) * KYN;
  const int row_off = ky0 - a.pad;   // input row of (output row y, local filter row ky) = y + row_off + ky
  const int item = blockIdx.y;                       // (image, column block, row segment)
  const int seg = item % a.segs, xb = (item / a.segs) % a.xblocks, n = item / (a.segs * a.xblocks);
  const int y_begin = seg * a.seg_rows, y_end = min(a.Ho, y_begin + a.seg_rows);
  const int xbase = xb * G3_PB;
  if (tid < NW * 16) {
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
  const u32x4 zero4 = {0u, 0u, 0u, 0u};

  // ---- staging map.  x: unit u = tid + NT k -> (pixel u / (2 NW), 8-channel chunk u % (2 NW)): 2 NW lanes cover one
  // pixel's bytes (NT is a multiple of 2 NW: the chunk is the same for every k).  dy: threads 0-255, (tid / 4, tid % 4).
  const int xchunk = tid % (2 * NW), xpix0 = tid / (2 * NW);
  constexpr int XPSTEP = C::NT / (2 * NW);   // 32
  const bool xc_ok = ci0 + xchunk * 8 < a.Cin;
  static_assert(C::NT % C::DPP == 0, "a thread's dy units share one 16-byte column");
  constexpr int DUNITS = G3_PB * C::DPP;          // dy units per row: 256 (NCO = 2) / 512 (NCO = 4)
  constexpr int DK = (DUNITS + C::NT - 1) / C::NT;   // dy units per thread (the last round partly idle)
  const int dpiece = tid % C::DPP;
  const bool dc_ok = co0 + dpiece * 8 < a.Cout;
  // The clamped address of a masked unit is the image's pixel 0, CHANNEL 0 -- the channel piece must be clamped with the pixel (it is
  // added inside the `ok ?` of the loads; as part of the base pointer it cost the 64-filter instantiation a spilled register): a piece past the last
  // channel (NW = 5 slices of an 64-channel input: chunks 8 and 9; filters past Cout) read at pixel 0 lies past the END of the tensor when
  // the image is the batch's last and a few pixels small (the legacy U-Nets' 1 x 1 / 2 x 2 stages).  That was round 5's intermittent
  // "Memory access fault by GPU": up to 32 bytes past a 512-byte tensor, harmless unless the caching allocator had put it at the very
  // end of a segment (found with tools/dbg/guard_alloc.cpp, GUARD_ALLOC_END=1 GUARD_ALLOC_LEAK=1: every tensor ends at an unmapped page).
  const int xcoff = ci0 + xchunk * 8, dcoff = co0 + dpiece * 8;      // added to the address only where the unit is real (ok implies xc_ok / dc_ok)
  const unsigned short* ximg = a.x + (long long)n * a.x_sn;
  const unsigned short* dimg = a.dy + (long long)n * a.dy_sn;
  int xdst[C::XK];
#pragma unroll
  for (int k = 0; k < C::XK; ++k) xdst[k] = C::xoff(xpix0 + XPSTEP * k, xchunk >> 1) + ((xchunk & 1) << 4);
  // two register sets: the rows of step y + 2 are requested while step y computes (round 5: with ONE set, requested at the top of a
  // step and stored at its end, every step waited out a whole memory latency -- 36 MFMAs of work against ~2 us -- MFMA busy 0.14)
  u32x4 xrA[C::XK], drA[DK], xrB[C::XK], drB[DK];
  unsigned xokA = 0, xokB = 0;
  // bias gradient = per-channel sum of dy: the workgroups of cin slice 0 / filter-row group 0 add up the dy rows they stage
  const bool want_bias = a.bias_part != nullptr && blockIdx.x == 0 && ky0 == 0;
  float bs[DK][8];
#pragma unroll
  for (int k = 0; k < DK; ++k)
#pragma unroll
    for (int e = 0; e < 8; ++e) bs[k][e] = 0.f;
  // bit k of xok: xr[k] holds raw data (else the unit is zero padding).  Loads are unconditional from a clamped address (a load
  // behind a branch is waited for at the join: DESIGN.md, status round 4 #7); the mask decides what the store writes.
  auto load_x_row = [&](int row, u32x4 (&xr)[C::XK], unsigned& xok) __attribute__((always_inline)) {
    xok = 0;
    const bool rok = row >= 0 && row < a.H && xc_ok;
#pragma unroll
    for (int k = 0; k < C::XK; ++k) {
      const int pix = xpix0 + XPSTEP * k, px = xbase - a.pad + pix;
      const bool ok = rok && px >= 0 && px < a.W && pix < C::XPIX;
      xr[k] = *reinterpret_cast<const u32x4*>(ximg + (ok ? (long long)row * a.x_sh + (long long)px * a.x_sw + xcoff : 0));
      xok |= ok ? 1u << k : 0u;
    }
  };
  auto store_x_row = [&](int row, const u32x4 (&xr)[C::XK], unsigned xok) __attribute__((always_inline)) {
    char* slot = Xs + ((unsigned)(row - row_off) % C::XSLOTS) * C::XROW_B;
#pragma unroll
    for (int k = 0; k < C::XK; ++k) {
      if (xpix0 + XPSTEP * k >= C::XPIX) break;
      u32x4 v = fd_xform8<FmtA, FmtG>(xr[k], sc_s + xchunk * 8, sh_s + xchunk * 8, a.pro_mode != 0 ? a.p_slope : 1.f);   // fp16 x -> bf16 operand
      v = ((xok >> k) & 1) ? v : zero4;   // zero padding of the ACTIVATED input (a select, not a branch)
      lds_write16(slot + xdst[k], v);
    }
  };
  auto load_d_row = [&](int row, u32x4 (&dr)[DK]) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < DK; ++k) {
      const int dpix = (tid + k * C::NT) / C::DPP;
      const bool ok = dpix < G3_PB && dc_ok && row < y_end && xbase + dpix < a.Wo;
      const u32x4 v = *reinterpret_cast<const u32x4*>(dimg + (ok ? (long long)row * a.dy_sh + (long long)(xbase + dpix) * a.dy_sw + dcoff : 0));
      dr[k] = ok ? v : zero4;
    }
  };
  auto store_d_row = [&](int row, const u32x4 (&dr)[DK]) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < DK; ++k) {
      const int dpix = (tid + k * C::NT) / C::DPP;
      if (dpix < G3_PB) lds_write16(Ds + (row & 1) * C::DROW_B + g3_doffn<NCO>(dpix, dpiece >> 1) + ((dpiece & 1) << 4), dr[k]);
      if (want_bias) {   // rows >= y_end were loaded as zeros
        const f32x8 f = __builtin_convertvector(__builtin_bit_cast(bf16x8, dr[k]), f32x8);
#pragma unroll
        for (int e = 0; e < 8; ++e) bs[k][e] += f[e];
      }
    }
  };

  // ---- fragment addresses: lane (g, i): k rows 8 g + (i >> 2) (+ 4 for the second read), 4-channel piece i & 3
  const int g = lane >> 4, i = lane & 15;
  const int kpix = 8 * g + (i >> 2), piece = (i & 3) * 8;
  f32x4 acc[C::TAPS][NCO];
#pragma unroll
  for (int t = 0; t < C::TAPS; ++t)
#pragma unroll
    for (int c = 0; c < NCO; ++c) acc[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nsub = (min(G3_PB, a.Wo - xbase) + 31) / 32;

  // input rows of the first output row, and its dy row
  for (int r = y_begin + row_off; r < y_begin + row_off + KYN; ++r) {
    load_x_row(r, xrA, xokA);
    store_x_row(r, xrA, xokA);
  }
  load_d_row(y_begin, drA);
  store_d_row(y_begin, drA);
  __syncthreads();
  // step y computes output row y from the staged rows, stores the rows of step y + 1 (requested in step y - 1, set A / B by parity)
  // and requests those of step y + 2
  // DEPTH 2 (two register sets) where the registers are there; the 9-wave 4x4 instantiation (168 registers per lane) keeps one set:
  // its rows are requested at the top of the step that stores them (DEPTH 1) -- still without a branch between request and store
  constexpr int DEPTH = (NW <= 8 && NCO == 2) ? 2 : 1;      // (64-filter workgroups: 144 accumulator registers, one staging set)
  if constexpr (DEPTH == 2) {
    load_x_row(y_begin + row_off + KYN, xrA, xokA);
    load_d_row(y_begin + 1, drA);
  }
  auto step = [&](int y, u32x4 (&xs)[C::XK], unsigned& xoks, u32x4 (&ds)[DK], u32x4 (&xn)[C::XK], unsigned& xokn, u32x4 (&dn)[DK]) __attribute__((always_inline)) {
    // NO control flow between these requests and the stores that consume them a step later: hipcc waits for a load at the end of
    // the conditional block it sits in (the first version had `if (!(a.dbg_skip & 4))` here and `vmcnt(0)` right behind the loads:
    // every step began by waiting out the latency of the rows it had just asked for -- MFMA busy 0.14)
    load_x_row(y + DEPTH - 1 + row_off + KYN, xn, xokn);   // in flight during this row's (DEPTH 2: AND the next row's) MFMAs
    load_d_row(y + DEPTH, dn);
    const char* dcur = Ds + (y & 1) * C::DROW_B;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      if (sub >= nsub) break;
      const int pa = 32 * sub + kpix;
      bf16x8 af[NCO];
#pragma unroll
      for (int c = 0; c < NCO; ++c) af[c] = g3_frag(dcur + g3_doffn<NCO>(pa, c) + piece, dcur + g3_doffn<NCO>(pa + 4, c) + piece);
#pragma unroll
      for (int ky = 0; ky < KYN; ++ky) {
        const char* slot = Xs + ((unsigned)(y + ky) % C::XSLOTS) * C::XROW_B;   // input row y + row_off + ky
#pragma unroll
        for (int kx = 0; kx < KS; ++kx) {
          const int pb = pa + kx;   // staged pixel index = output pixel + kx (staged pixel 0 is x = xbase - pad)
          const bf16x8 bfr = g3_frag(slot + C::xoff(pb, wave) + piece, slot + C::xoff(pb + 4, wave) + piece);
#pragma unroll
          for (int c = 0; c < NCO; ++c)
            acc[ky * KS + kx][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[c], bfr, acc[ky * KS + kx][c], 0, 0, 0);
        }
      }
    }
    store_x_row(y + row_off + KYN, xs, xoks);   // its slot held the row above this step's first: last read one barrier ago
    store_d_row(y + 1, ds);
    // LDS-only barrier: __syncthreads() carries s_waitcnt vmcnt(0), which would wait for the rows just requested for step y + 2
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };
  // (both steps unconditionally: an `if (y + 1 < y_end)` around the second one is a conditional block with loads in it, drained at its
  // end.  A step past the segment multiplies a dy row that was loaded as zeros -- `row < y_end` in load_d_row -- and adds nothing.)
  if constexpr (DEPTH == 2) {
    for (int y = y_begin; y < y_end; y += 2) {
      step(y, xrA, xokA, drA, xrB, xokB, drB);
      step(y + 1, xrB, xokB, drB, xrA, xokA, drA);
    }
  } else {
    for (int y = y_begin; y < y_end; ++y) step(y, xrA, xokA, drA, xrA, xokA, drA);
  }
  if (want_bias) {   // after the loop's last barrier: the x slots are free
    constexpr int BC = NCO * 16;                 // this workgroup's filters
    float* red = reinterpret_cast<float*>(Xs);   // [64 pixels][BC channels]
#pragma unroll
    for (int k = 0; k < DK; ++k) {
      const int dpix = (tid + k * C::NT) / C::DPP;
      if (dpix < G3_PB)
#pragma unroll
        for (int e = 0; e < 8; ++e) red[dpix * BC + dpiece * 8 + e] = bs[k][e];
    }
    __syncthreads();
    if (tid < BC) {
      float t = 0.f;
      for (int q = 0; q < G3_PB; ++q) t += red[q * BC + tid];
      // [item][32-filter group][32]: this workgroup's NCO / 2 groups are consecutive (groups past the last filter are never read)
      const long long grp = (long long)(blockIdx.z / KYG) * (NCO / 2) + (tid >> 5);
      if (grp < a.zt / KYG) a.bias_part[((long long)item * (a.zt / KYG) + grp) * 32 + (tid & 31)] = t;
    }
  }
  // partial sums in accumulator order, [item][cin slice][z][wave][tap][cout tile][r][lane]: every store is 256
  // contiguous bytes (scattering them to [co][ci][tap] here cost 50 us per launch); wgrad_tr_reduce_kernel maps back
  if (!(a.dbg_skip & 1)) {
    // the layout stays [item][cin slice][z: 32-filter group x filter-row group][wave][tap][2 cout tiles][r][lane]: a 64-filter
    // workgroup writes its two halves as z = 2 blockIdx.z and 2 blockIdx.z + 1 (a.zt groups in all; the half past the last is skipped)
#pragma unroll
    for (int h = 0; h < NCO / 2; ++h) {
      const int z = (int)blockIdx.z * (NCO / 2) + h;
      if (z >= a.zt) break;
      float* blk = a.part + ((((long long)item * gridDim.x + blockIdx.x) * a.zt + z) * NW + wave) * (C::TAPS * 512) + lane;
#pragma unroll
      for (int t = 0; t < C::TAPS; ++t)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
          for (int r = 0; r < 4; ++r) blk[((t * 2 + ct) * 4 + r) * 64] = acc[t][h * 2 + ct][r];
    }
  }
}

// ---- second generation, 3x3 pad 1 without a bias: walk INPUT rows, keep the dy fragments of three output rows in registers.
// Input row r serves filter row ky of output row r - ky + 1, so one B fragment (row r, shift kx) feeds 3 filter rows x 2
// cout tiles = 6 MFMAs instead of 2, and a step reads 10 fragments instead of 22 (the first kernel sat at 13 % MFMA busy:
// every wave waited on its 44 transpose reads per row, then on the staging, then on the barrier).  A step is software-
// pipelined inside each wave:
//   * the fragments of row r + 1 are requested while row r's MFMAs run -- every B fragment is refilled in place right after
//     the block of 6 MFMAs that used it, the newest dy fragments after the last block of their 32-pixel half;
//   * rows travel HBM -> LDS by LDS-DMA (global_load_lds_dwordx4, no staging registers), R3_PFX + 2 rows ahead: raw input
//     rows into a ring of 6 linear slots, dy rows straight into their fragment layout (the swizzle is applied to the SOURCE
//     address, the LDS side of the DMA is lane-linear).  A wave transforms the units it fetched itself (BatchNorm + ReLU +
//     zero padding, raw slot -> one of two swizzled slots), two dwords per MFMA block, so its own s_waitcnt vmcnt is the only
//     synchronisation between fetch and transform;
//   * one raw barrier per row (s_waitcnt lgkmcnt(0); s_barrier -- __syncthreads() would also drain the DMA queue).
// Six steps are unrolled so that every ring slot and register set is a compile-time constant.  Same work split and the same
// partial-sum layout as the first kernel (wgrad_tr_reduce_kernel maps it back): the summation order is unchanged.
template <int V> struct IC { static constexpr int value = V; };
#define R3_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
// The step's barrier: LDS operations of a wave complete in order, and the last six issued before it are the fragment reads
// that refill B[1][2] and the newest dy fragments of the second half -- wanted only at the END of the next step.  Waiting for
// all but those six publishes this wave's transformed units without exposing the reads' latency at every barrier.
#define R3_STEP_BARRIER() asm volatile("s_waitcnt lgkmcnt(6)\n\ts_barrier" ::: "memory")
constexpr int R3_PFX = 4;                       // group m (input row m + dy row m) is requested in step m - 2 - R3_PFX
constexpr int R3_RAW_B = 66 * 256, R3_D_B = 64 * 64;
constexpr int R3_LDS = 6 * R3_RAW_B + 2 * R3_RAW_B + 6 * R3_D_B + 1024;

// LDS-DMA of 16 bytes per lane: global (uniform base + 32-bit lane byte offset) -> LDS (uniform base + 16 lane).  Inline
// assembly on purpose: with the builtin hipcc knows the instruction writes LDS and puts s_waitcnt vmcnt(0) in front of the
// wave's next LDS read, i.e. drains the whole prefetch queue every step.  The waits are counted by hand (r3_wait_vm).
__device__ __forceinline__ void r3_dma16(const unsigned short* base, unsigned lane_bytes, char* lds_wave_base) {
  const unsigned lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds_wave_base;
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lane_bytes), "s"(base), "s"(lds) : "memory", "m0");
}
template <int N>
__device__ __forceinline__ void r3_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// one dword (two channels) of the prologue transform; RELU: BatchNorm + ReLU on the packed result (fd_bn_relu8's recipe)
template <bool RELU>
__device__ __forceinline__ unsigned r3_xform2(unsigned raw, float sc0, float sc1, float sh0, float sh1, float slope, unsigned mask) {
  typedef f32x2 f32x2_t;
  typedef __attribute__((ext_vector_type(2))) short s16x2_t;
  f32x2_t f = fd_cvt2<FmtA>(raw);   // the forward input is fp16; the result is the bf16 operand multiplied with dy
  f = __builtin_elementwise_fma(f, (f32x2_t){sc0, sc1}, (f32x2_t){sh0, sh1});
  unsigned out;
  if constexpr (RELU) {
    const s16x2_t pk = __builtin_bit_cast(s16x2_t, fd_pk2<FmtG>(f));
    out = __builtin_bit_cast(unsigned, __builtin_elementwise_max(pk, (s16x2_t){0, 0}));
  } else {
    f = __builtin_elementwise_max(f, f * slope);
    out = fd_pk2<FmtG>(f);
  }
  return out & mask;   // zero padding of the ACTIVATED input
}

// Cin a multiple of 128 (one 128-channel slice per workgroup), Cout a multiple of 32, W a multiple of 64
// DBG (tuning builds only, results wrong): 1 no MFMAs, 2 no transform arithmetic, 4 no DMA requests / waits, 8 no fragment refills
template <bool RELU, int DBG = 0>
__global__ __launch_bounds__(512) void conv_wgrad_r3_kernel(WgradRowsArgs a) {
  using C = G3Cfg<3, 3, 8>;
  static_assert(C::XROW_B == R3_RAW_B && C::DROW_B == R3_D_B, "slot sizes");
  extern __shared__ __attribute__((aligned(16))) char g3_lds[];
  char* Raw = g3_lds;                               // 6 raw input rows, linear: unit u = pixel u / 16, 8-channel chunk u % 16
  char* Xs = g3_lds + 6 * R3_RAW_B;                 // 2 transformed rows in the fragment layout
  char* Ds = Xs + 2 * R3_RAW_B;                     // 6 dy rows in the fragment layout
  float* sc_s = reinterpret_cast<float*>(Ds + 6 * R3_D_B);   // scale / shift until they are in registers, then the DMA dump
  float* sh_s = sc_s + 128;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ci0 = blockIdx.x * 128, co0 = (int)blockIdx.z * 32;
  const int item = blockIdx.y;
  const int seg = item % a.segs, xb = (item / a.segs) % a.xblocks, n = item / (a.segs * a.xblocks);
  const int y_begin = seg * a.seg_rows, y_end = min(a.Ho, y_begin + a.seg_rows);
  const int nsteps = y_end - y_begin + 2;     // input rows y_begin - 1 .. y_end
  const int r0 = y_begin - 1, xbase = xb * G3_PB;
  if (tid < 128) {
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
  const float slope = a.pro_mode == 0 ? 1.f : a.p_slope;   // no prologue: max(t, 1 t) = t
  // ---- a thread's units: input pixels tid / 16 and 32 + tid / 16 (chunk tid % 16) of every row, fetched by its wave's two
  // full DMA instructions; the halo pixels 64, 65 are wave 0's third instruction (32 lanes); dy rows: 32 lanes of every wave.
  const int xchunk = tid & 15, xpix0 = tid >> 4;
  const f32x4 s0 = *reinterpret_cast<const f32x4*>(sc_s + xchunk * 8), s1 = *reinterpret_cast<const f32x4*>(sc_s + xchunk * 8 + 4);
  const f32x4 h0 = *reinterpret_cast<const f32x4*>(sh_s + xchunk * 8), h1 = *reinterpret_cast<const f32x4*>(sh_s + xchunk * 8 + 4);
  __syncthreads();   // every thread holds its scale / shift: the area becomes the dump of the other waves' third DMA
  // global sources as a wave-uniform row pointer + a 32-bit lane offset (saddr-form DMA: no 64-bit lane arithmetic)
  const unsigned short* ximg = a.x + (long long)n * a.x_sn + ci0;
  char* xwp[3];            // where this thread's transformed units go (slot 0)
  unsigned xsrc[3];
  unsigned xcol[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int pix = xpix0 + 32 * k, px = xbase - 1 + pix;
    xwp[k] = Xs + C::xoff(pix, xchunk >> 1) + ((xchunk & 1) << 4);
    xsrc[k] = 2u * (unsigned)(min(max(px, 0), a.W - 1) * a.x_sw + xchunk * 8);   // bytes
    xcol[k] = px >= 0 && px < a.W ? 0xffffffffu : 0u;
  }
  const bool halo_t = tid < 32;
  // dy row: waves 1-4, lane l of wave w fetches linear position q = 64 (w - 1) + l of the row's fragment layout -> pixel q / 4,
  // 16-filter group ((q >> 1) & 1) ^ ((pixel >> 3) & 1), half q & 1
  const int dq = ((wave - 1) & 3) * 64 + lane, dpix = dq >> 2;
  const unsigned short* dimg = a.dy + (long long)n * a.dy_sn + co0;
  const unsigned dsrc = 2u * (unsigned)((xbase + dpix) * a.dy_sw + ((((dq >> 1) & 1) ^ ((dpix >> 3) & 1)) << 4) + ((dq & 1) << 3));
  char* dump = reinterpret_cast<char*>(sc_s);
  // group m: input row r0 + m (clamped: rows outside the image are masked by the transform) and dy row y_begin + m (rows
  // outside the segment: zeros are written instead, the DMA goes to the dump so that the instruction count stays the same).
  // Waves 0-4 issue three DMA instructions per group (wave 0: the halo pixels, 1-4: a quarter of the dy row), waves 5-7 two.
  const bool three = wave <= 4;
  auto request = [&](int m, auto SLOT) __attribute__((always_inline)) {
    constexpr int sl = decltype(SLOT)::value;
    if constexpr (DBG & 4) return;
    const unsigned short* rp = ximg + (long long)min(max(r0 + m, 0), a.H - 1) * a.x_sh;
    char* raw = Raw + sl * R3_RAW_B + wave * 1024;
    r3_dma16(rp, xsrc[0], raw);
    r3_dma16(rp, xsrc[1], raw + 8192);
    if (wave == 0) {
      if (lane < 32) r3_dma16(rp, xsrc[2], Raw + sl * R3_RAW_B + 16384);
    } else if (three) {
      char* dslot = Ds + sl * R3_D_B + (wave - 1) * 1024;
      if (y_begin + m < y_end) {
        r3_dma16(dimg + (long long)(y_begin + m) * a.dy_sh, dsrc, dslot);
      } else {
        if (lane == 0) r3_dma16(rp, xsrc[0], dump + wave * 16);
        lds_write16(dslot + lane * 16, u32x4{0u, 0u, 0u, 0u});
      }
    }
  };
  auto wait_landed = [&](auto NG) __attribute__((always_inline)) {   // at most NG younger groups of this wave still in flight
    constexpr int ng = decltype(NG)::value;
    if constexpr (DBG & 4) return;
    if (three) r3_wait_vm<3 * ng>(); else r3_wait_vm<2 * ng>();
  };
  auto rowmask = [&](int m) __attribute__((always_inline)) -> unsigned {   // is input row r0 + m inside the image (and wanted)
    const int row = r0 + m;
    return m < nsteps && row >= 0 && row < a.H ? 0xffffffffu : 0u;
  };
  // dword q of a unit through the prologue, masked
  auto xf = [&](u32x4& v, int q, unsigned m) __attribute__((always_inline)) {
    const float sa = q < 2 ? s0[2 * q] : s1[2 * q - 4], sb = q < 2 ? s0[2 * q + 1] : s1[2 * q - 3];
    const float ha = q < 2 ? h0[2 * q] : h1[2 * q - 4], hb = q < 2 ? h0[2 * q + 1] : h1[2 * q - 3];
    if constexpr (!(DBG & 2)) v[q] = r3_xform2<RELU>(v[q], sa, sb, ha, hb, slope, m);
  };
  auto xform_halo = [&](unsigned rm, auto RS, auto XS) __attribute__((always_inline)) {   // pixels 64, 65: threads 0-31
    if constexpr (DBG & 64) return;
    if (halo_t) {
      u32x4 v = *reinterpret_cast<const u32x4*>(Raw + decltype(RS)::value * R3_RAW_B + 16384 + tid * 16);
#pragma unroll
      for (int q = 0; q < 4; ++q) xf(v, q, rm & xcol[2]);
      lds_write16(xwp[2] + decltype(XS)::value * R3_RAW_B, v);
    }
  };

  // fragment offsets inside a row slot: lane (g, i) -> k rows 8 g + (i >> 2) (+ 4 for the second read), piece i & 3;
  // the second 32-pixel half of a row is a constant 8192 / 2048 bytes further (the swizzles repeat every 16 pixels)
  const int g = lane >> 4, i = lane & 15;
  const int kpix = 8 * g + (i >> 2), piece = (i & 3) * 8;
  // per-lane POINTERS into slot 0, so that slot and half are immediate offsets of the reads
  const char *bp[3][2], *ap[2][2];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) {
    bp[kx][0] = Xs + C::xoff(kpix + kx, wave) + piece;
    bp[kx][1] = Xs + C::xoff(kpix + kx + 4, wave) + piece;
  }
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    ap[ct][0] = Ds + g3_doff(kpix, ct) + piece;
    ap[ct][1] = Ds + g3_doff(kpix + 4, ct) + piece;
  }
  const char* rawp = Raw + tid * 16;
  f32x4 acc[9][2];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t][0] = acc[t][1] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 A[3][2][2], B[2][3];
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  const bf16x8 zfrag = __builtin_bit_cast(bf16x8, zero4);
#pragma unroll
  for (int s = 0; s < 3; ++s)
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) A[s][sub][0] = A[s][sub][1] = zfrag;

  // prologue: groups 0 .. R3_PFX + 1 requested; rows 0 and 1 transformed; the fragments of row 0 read
  request(0, IC<0>{});
  request(1, IC<1>{});
  request(2, IC<2>{});
  request(3, IC<3>{});
  request(4, IC<4>{});
  request(5, IC<5>{});
  static_assert(R3_PFX == 4, "six groups in flight = six ring slots");
  wait_landed(IC<R3_PFX>{});
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const unsigned rm = rowmask(j);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      u32x4 v = *reinterpret_cast<const u32x4*>(rawp + j * R3_RAW_B + k * 8192);
#pragma unroll
      for (int q = 0; q < 4; ++q) xf(v, q, rm & xcol[k]);
      lds_write16(xwp[k] + j * R3_RAW_B, v);
    }
    if (j == 0) xform_halo(rm, IC<0>{}, IC<0>{}); else xform_halo(rm, IC<1>{}, IC<1>{});
  }
  R3_BARRIER();
#pragma unroll
  for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) B[sub][kx] = g3_frag(bp[kx][0] + sub * 8192, bp[kx][1] + sub * 8192);
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) A[0][sub][ct] = g3_frag(ap[ct][0] + sub * 2048, ap[ct][1] + sub * 2048);
  }
  // Every wave's first fragment reads (row 0: transformed slot 0, dy slot 0) must have COMPLETED before any wave enters step 0,
  // which rewrites those very slots (row 2 into transformed slot 0 a third of the way in, the DMA of row 6 into dy slot 0): in
  // the steady state a step's barrier separates the reads of a slot from its next writer, here nothing did.  A wave held back
  // right after the barrier above -- e.g. by a co-resident kernel's s_setprio'd waves, which is what the training step's
  // second stream puts beside this kernel -- then read the NEXT rows' data (round 3: 1 % errors that came and went with what
  // ran on the other stream; invisible in every single-stream test).
  R3_BARRIER();

  // step j (phase P = j mod 6): MFMAs of input row j; dy row j - ky is in A[(j - ky) mod 3].  Six blocks of 6 MFMAs; the
  // transform of row j + 2's two full units is spread over blocks 1-4 (two dwords each), pinned by sched_barriers so that
  // neither the reads nor the VALU work pile up in front of the MFMAs.
  auto step = [&](int j, auto P) __attribute__((always_inline)) {
    constexpr int p = decltype(P)::value, p3 = p % 3, p2 = p & 1;
    constexpr int xn = (p2 ^ 1) * R3_RAW_B;             // transformed slot of row j + 1
    constexpr int dn = ((p + 1) % 6) * R3_D_B;          // dy row j + 1
    constexpr int xw = p2 * R3_RAW_B;                   // transformed slot of row j = of row j + 2
    constexpr int rw = ((p + 2) % 6) * R3_RAW_B;        // raw slot of row j + 2
    const unsigned rm = rowmask(j + 2);
    wait_landed(IC<R3_PFX - 1>{});                      // group j + 2 has landed (this wave's part of it)
    u32x4 u[2];
    u[0] = *reinterpret_cast<const u32x4*>(rawp + rw);
    u[1] = *reinterpret_cast<const u32x4*>(rawp + rw + 8192);
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int blk = sub * 3 + kx;
#pragma unroll
        for (int ky = 2; ky >= 0; --ky) {
          const int s = (p3 + 3 - ky) % 3;
          if constexpr (DBG & 1) {
            acc[ky * 3 + kx][0][0] += (float)A[s][sub][0][0] * (float)B[sub][kx][0];
            acc[ky * 3 + kx][1][0] += (float)A[s][sub][1][0] * (float)B[sub][kx][0];
            continue;
          }
          acc[ky * 3 + kx][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[s][sub][0], B[sub][kx], acc[ky * 3 + kx][0], 0, 0, 0);
          acc[ky * 3 + kx][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[s][sub][1], B[sub][kx], acc[ky * 3 + kx][1], 0, 0, 0);
        }
        if constexpr (!(DBG & 8)) B[sub][kx] = g3_frag(bp[kx][0] + (xn + sub * 8192), bp[kx][1] + (xn + sub * 8192));
        if (blk >= 1 && blk <= 4) {
          const int k = (blk - 1) >> 1, q0 = ((blk - 1) & 1) * 2;
          xf(u[k], q0, rm & xcol[k]);
          xf(u[k], q0 + 1, rm & xcol[k]);
          if ((blk - 1) & 1) lds_write16(xwp[k] + xw, u[k]);
        }
        if (kx == 2 && !(DBG & 8)) {
#pragma unroll
          for (int ct = 0; ct < 2; ++ct) A[(p3 + 1) % 3][sub][ct] = g3_frag(ap[ct][0] + (dn + sub * 2048), ap[ct][1] + (dn + sub * 2048));
        }
        // a wave issues in order: six MFMAs in a row would hold the block's VALU work back for ~100 cycles while the
        // matrix pipe drains them one by one.  Two VALU instructions ride in every MFMA's shadow instead.
#pragma unroll
        for (int q = 0; q < 6; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (blk == 4) {   // before the last block: its refills stay the youngest LDS operations of the step
          xform_halo(rm, IC<(p + 2) % 6>{}, IC<p2>{});
          request(j + 2 + R3_PFX, IC<p>{});            // into the ring slots of row j (raw: transformed two steps ago)
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    if constexpr (DBG & 32) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
    else if constexpr (DBG & 8) R3_BARRIER(); else R3_STEP_BARRIER();
  };
  // whole groups of six steps (one exit: with a break after every step the accumulators lost their fixed registers);
  // steps past the segment's last row see zero rows.  Row counts of 4, 16 and 64 per segment give 6, 18 and 66 steps.
  for (int j = 0; j < nsteps; j += 6) {
    step(j, IC<0>{});
    step(j + 1, IC<1>{});
    step(j + 2, IC<2>{});
    step(j + 3, IC<3>{});
    step(j + 4, IC<4>{});
    step(j + 5, IC<5>{});
  }
  r3_wait_vm<0>();   // the last requests land in LDS nobody reads; do not leave them in flight at exit
  float* blk = a.part + ((((long long)item * gridDim.x + blockIdx.x) * gridDim.z + blockIdx.z) * 8 + wave) * (9 * 512) + lane;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) blk[((t * 2 + ct) * 4 + r) * 64] = acc[t][ct][r];
}

// ---- the same scheme for the Fusion-discriminator's MFMA-bound 4x4 (stride 1, pad 1, Cin = 144, /root/reference/models/
// dehaze1113.py:214): 16 taps, so one B fragment feeds 4 filter rows x 2 cout tiles = 8 MFMAs and the dy fragments of FOUR
// output rows rotate through registers (6 fragment reads per 32 MFMAs; the first kernel: 10 per 16, two filter rows per
// workgroup, every input row staged 18 times).  Workgroup = 3 waves = one 48-channel slice of Cin (144 = 3 slices, no
// padding), 32-pixel column blocks (the 128 accumulator registers of 16 taps leave room for one 32-pixel half only), two
// workgroups per CU.  Rings of 4 (input rows raw, dy rows), 4 steps unrolled.  dy pixels past the last output column are
// never fetched: their LDS bytes keep the zeros written once.
constexpr int R4_PX = 32, R4_XPIX = 35;                   // output pixels per block, staged input pixels
constexpr int R4_RAW_B = 3456;                            // raw row: 35 px x 96 B = 3360, rounded up to whole 1 KiB DMA pieces + tail
constexpr int R4_X_B = R4_XPIX * 96 + (R4_XPIX / 8 + 1) * 128;   // transformed row in the odd-tile layout of G3Cfg<.,.,3>
constexpr int R4_D_B = R4_PX * 64;
constexpr int R4_LDS = 4 * R4_RAW_B + 2 * R4_X_B + 4 * R4_D_B + 512 + 192 * 32;
#define R4_STEP_BARRIER() asm volatile("s_waitcnt lgkmcnt(6)\n\ts_barrier" ::: "memory")

template <bool RELU>
__global__ __launch_bounds__(192, 2) void conv_wgrad_r4_kernel(WgradRowsArgs a) {
  using C = G3Cfg<4, 2, 3>;       // only its layout function xoff() (odd number of tiles: shifted blocks)
  extern __shared__ __attribute__((aligned(16))) char g3_lds[];
  char* Raw = g3_lds;                               // 4 raw rows, linear: unit u = pixel u / 6, 8-channel chunk u % 6
  char* Xs = Raw + 4 * R4_RAW_B;                    // 2 transformed rows
  char* Ds = Xs + 2 * R4_X_B;                       // 4 dy rows in fragment layout
  float* sc_s = reinterpret_cast<float*>(Ds + 4 * R4_D_B);
  float* sh_s = sc_s + 48;
  // per-thread loop invariants used once per step, as [value][thread] (one record per thread was a 16-way bank conflict per read)
  unsigned* invb = reinterpret_cast<unsigned*>(Ds + 4 * R4_D_B + 512) + threadIdx.x;
#define inv(k) invb[(k) * 192]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD-aware work numbering.  The 27 workgroups of one item (3 cin tiles x 9 cout slices for D's 144 -> 288) read the SAME x and
  // dy rows; as a (3, items, 9) grid they were dispatched round-robin over the 8 XCDs and far apart in time, and every one of
  // them pulled its rows from HBM: rocprofv3 FETCH_SIZE 2.29 GB per launch for 224 MB of input (round 2's PMC file).  Launched
  // 1-D, workgroup L goes to XCD L % 8 and is that XCD's (L / 8)-th: consecutive slots of one XCD walk the 27 pieces of one
  // item, so the rows are fetched once into that XCD's L2 and hit there 26 times.
  const int vgx = a.vgx, vgz = a.vgz;
  int bx, bz, item;
  {
    const int L = (int)blockIdx.x, per = vgx * vgz, slot = L >> 3;
    const int it8 = slot / per;
    int sub = slot - it8 * per;
    item = it8 * 8 + (L & 7);
    if (a.dbg_skip & 256) {                          // tuning aid: the plain (cin tile, item, cout slice) order of a 3-D grid
      sub = (L % vgx) + vgx * (L / (vgx * ((a.vgy + 7) / 8 * 8)));
      item = (L / vgx) % ((a.vgy + 7) / 8 * 8);
    }
    if (item >= a.vgy) return;                       // padding of the last group of 8 items (whole workgroup)
    bx = sub % vgx, bz = sub / vgx;
  }
  const int ci0 = bx * 48, co0 = bz * 32;
  const int seg = item % a.segs, xb = (item / a.segs) % a.xblocks, n = item / (a.segs * a.xblocks);
  const int y_begin = seg * a.seg_rows, y_end = min(a.Ho, y_begin + a.seg_rows);
  const int nsteps = y_end - y_begin + 3;          // input rows y_begin - 1 .. y_end + 1
  const int r0 = y_begin - a.pad, xbase = xb * R4_PX;
  if (tid < 48) {
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
  for (int i = tid; i < 4 * R4_D_B / 16; i += 192) lds_write16(Ds + i * 16, u32x4{0u, 0u, 0u, 0u});   // never-fetched dy pixels stay zero
  __syncthreads();
  const float slope = a.pro_mode == 0 ? 1.f : a.p_slope;
  // units: thread -> input pixel tid / 6 (0 .. 31), chunk tid % 6; pixels 32 .. 34 are threads 0-17's second unit
  const int xchunk = tid % 6, xpix0 = tid / 6;
  // scale / shift of the thread's 8 channels stay in LDS (two floats of each per transformed dword): 16 resident registers
  // more made hipcc spill pointers, and a scratch reload drains the DMA queue
  const float* scp = sc_s + xchunk * 8;
  const float* shp = sh_s + xchunk * 8;
  const unsigned short* ximg = a.x + (long long)n * a.x_sn + ci0;
  // the main unit's addresses stay in registers; the tail unit's and the DMA source offsets are read back from LDS where they
  // are used (once per step each): in registers they were what hipcc spilled, and a scratch reload is a vmcnt(0)
  char* xwp0;
  unsigned xcol0;
  {
    const int px = xbase - a.pad + xpix0;
    xwp0 = Xs + C::xoff(xpix0, xchunk >> 1) + ((xchunk & 1) << 4);
    xcol0 = px >= 0 && px < a.W ? 0xffffffffu : 0u;
    const int pix1 = min(xpix0 + 32, R4_XPIX - 1), px1 = xbase - a.pad + xpix0 + 32;
    inv(0) = (unsigned)(C::xoff(pix1, xchunk >> 1) + ((xchunk & 1) << 4));                       // tail unit: offset in a transformed slot
    inv(1) = px1 >= 0 && px1 < a.W ? 0xffffffffu : 0u;                                            //            column mask
    inv(2) = 2u * (unsigned)(min(max(px, 0), a.W - 1) * a.x_sw + xchunk * 8);                     // DMA source offsets (bytes)
    inv(3) = 2u * (unsigned)(min(max(px1, 0), a.W - 1) * a.x_sw + xchunk * 8);
  }
  const bool tail_t = tid < 18;
  // dy row: waves 1, 2; lane l of wave w fetches position q = 64 (w - 1) + l of the row's fragment layout
  const int dq = ((wave + 1) & 1) * 64 + lane, dpix = dq >> 2;          // wave 1 -> 0 .. 63, wave 2 -> 64 .. 127
  const bool d_lane = xbase + dpix < a.Wo;
  const unsigned short* dimg = a.dy + (long long)n * a.dy_sn + co0;
  inv(4) = 2u * (unsigned)(min(xbase + dpix, a.Wo - 1) * a.dy_sw + ((((dq >> 1) & 1) ^ ((dpix >> 3) & 1)) << 4) + ((dq & 1) << 3));
  char* dump = reinterpret_cast<char*>(sc_s) + 384;      // 128 bytes behind the 96 floats of scale / shift
  // group m: input row r0 + m and dy row y_begin + m; exactly two DMA instructions per wave (wave 0: 64 units + the 18-unit
  // tail, waves 1 and 2: 64 units + half a dy row)
  auto request = [&](int m, auto SLOT) __attribute__((always_inline)) {
    constexpr int sl = decltype(SLOT)::value;
    const unsigned short* rp = ximg + (long long)min(max(r0 + m, 0), a.H - 1) * a.x_sh;
    const unsigned xsrc0 = inv(2);
    r3_dma16(rp, xsrc0, Raw + sl * R4_RAW_B + wave * 1024);
    if (wave == 0) {
      if (lane < 18) r3_dma16(rp, inv(3), Raw + sl * R4_RAW_B + 3072);
    } else {
      char* dslot = Ds + sl * R4_D_B + (wave - 1) * 1024;
      if (y_begin + m < y_end) {
        if (d_lane) r3_dma16(dimg + (long long)(y_begin + m) * a.dy_sh, inv(4), dslot);
      } else {
        if (lane == 0) r3_dma16(rp, xsrc0, dump + wave * 16);
        unsigned z;      // materialised here: hoisted out of the loop, the four zero registers were spilled and reloaded every step
        asm volatile("v_mov_b32 %0, 0" : "=v"(z));
        lds_write16(dslot + lane * 16, u32x4{z, z, z, z});
      }
    }
  };
  auto rowmask = [&](int m) __attribute__((always_inline)) -> unsigned {
    const int row = r0 + m;
    return m < nsteps && row >= 0 && row < a.H ? 0xffffffffu : 0u;
  };
  typedef __attribute__((ext_vector_type(2))) float f2_t;
  f2_t sv = *reinterpret_cast<const f2_t*>(scp), hv = *reinterpret_cast<const f2_t*>(shp);   // the pair of the next dword to transform,
                                                                                            // read one MFMA block ahead (wraps to dword 0)
  auto xf = [&](u32x4& v, int q, unsigned m) __attribute__((always_inline)) {
    const f2_t sv = *reinterpret_cast<const f2_t*>(scp + 2 * q), hv = *reinterpret_cast<const f2_t*>(shp + 2 * q);
    v[q] = r3_xform2<RELU>(v[q], sv[0], sv[1], hv[0], hv[1], slope, m);
  };
  auto xform_tail = [&](unsigned rm, auto RS, auto XS) __attribute__((always_inline)) {   // pixels 32 .. 34: threads 0-17
    if (tail_t) {
      u32x4 v = *reinterpret_cast<const u32x4*>(Raw + decltype(RS)::value * R4_RAW_B + 3072 + tid * 16);
#pragma unroll
      for (int q = 0; q < 4; ++q) xf(v, q, rm & inv(1));
      lds_write16(Xs + inv(0) + decltype(XS)::value * R4_X_B, v);
    }
  };
  const int g = lane >> 4, i = lane & 15;
  const int kpix = 8 * g + (i >> 2), piece = (i & 3) * 8;
  const char *bp[4][2], *ap[2][2];
#pragma unroll
  for (int kx = 0; kx < 4; ++kx) {
    bp[kx][0] = Xs + C::xoff(kpix + kx, wave) + piece;
    bp[kx][1] = Xs + C::xoff(kpix + kx + 4, wave) + piece;
  }
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    ap[ct][0] = Ds + g3_doff(kpix, ct) + piece;
    ap[ct][1] = Ds + g3_doff(kpix + 4, ct) + piece;
  }
  const char* rawp = Raw + tid * 16;
  f32x4 acc[16][2];
#pragma unroll
  for (int t = 0; t < 16; ++t) acc[t][0] = acc[t][1] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 A[4][2], B[4];
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  const bf16x8 zfrag = __builtin_bit_cast(bf16x8, zero4);
#pragma unroll
  for (int s = 0; s < 4; ++s) A[s][0] = A[s][1] = zfrag;

  request(0, IC<0>{});
  request(1, IC<1>{});
  request(2, IC<2>{});
  request(3, IC<3>{});
  r3_wait_vm<4>();       // groups 0 and 1 have landed (two instructions per group and wave)
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const unsigned rm = rowmask(j);
    u32x4 v = *reinterpret_cast<const u32x4*>(rawp + j * R4_RAW_B);
#pragma unroll
    for (int q = 0; q < 4; ++q) xf(v, q, rm & xcol0);
    lds_write16(xwp0 + j * R4_X_B, v);
    if (j == 0) xform_tail(rm, IC<0>{}, IC<0>{}); else xform_tail(rm, IC<1>{}, IC<1>{});
  }
  R3_BARRIER();
#pragma unroll
  for (int kx = 0; kx < 4; ++kx) B[kx] = g3_frag(bp[kx][0], bp[kx][1]);
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) A[0][ct] = g3_frag(ap[ct][0], ap[ct][1]);
  R3_BARRIER();      // as in conv_wgrad_r3_kernel: the first fragment reads of every wave complete before step 0 rewrites their slots

  // step j (phase P = j mod 4): MFMAs of input row j; dy row j - ky is in A[(j - ky) mod 4]
  auto step = [&](int j, auto P) __attribute__((always_inline)) {
    constexpr int p = decltype(P)::value, p2 = p & 1;
    constexpr int xn = (p2 ^ 1) * R4_X_B, dn = ((p + 1) & 3) * R4_D_B, xw = p2 * R4_X_B, rw = ((p + 2) & 3) * R4_RAW_B;
    const unsigned rm = rowmask(j + 2);
    r3_wait_vm<2>();                                   // group j + 2 has landed (this wave's part)
    u32x4 u = *reinterpret_cast<const u32x4*>(rawp + rw);
#pragma unroll
    for (int kx = 0; kx < 4; ++kx) {
      const f2_t svn = *reinterpret_cast<const f2_t*>(scp + 2 * ((kx + 1) & 3)), hvn = *reinterpret_cast<const f2_t*>(shp + 2 * ((kx + 1) & 3));
#pragma unroll
      for (int ky = 3; ky >= 0; --ky) {
        const int s = (p + 4 - ky) & 3;
        acc[ky * 4 + kx][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[s][0], B[kx], acc[ky * 4 + kx][0], 0, 0, 0);
        acc[ky * 4 + kx][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[s][1], B[kx], acc[ky * 4 + kx][1], 0, 0, 0);
      }
      u[kx] = r3_xform2<RELU>(u[kx], sv[0], sv[1], hv[0], hv[1], slope, rm & xcol0);
      sv = svn, hv = hvn;
      // the step's LDS WRITE goes out before the last block's fragment refills: R4_STEP_BARRIER waits for everything but the
      // six youngest LDS operations, which must all be reads (with the write after B[3]'s refill it sat among those six and
      // was not covered by the barrier: the D 4x4 weight gradient differed from run to run at 127 x 127, round 3)
      if (kx == 3) lds_write16(xwp0 + xw, u);
      B[kx] = g3_frag(bp[kx][0] + xn, bp[kx][1] + xn);
      if (kx == 3) {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) A[(p + 1) & 3][ct] = g3_frag(ap[ct][0] + dn, ap[ct][1] + dn);
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (kx == 2) {   // before the last block: its refills stay the youngest LDS operations of the step
        xform_tail(rm, IC<(p + 2) & 3>{}, IC<p2>{});
        request(j + 4, IC<p>{});
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    R4_STEP_BARRIER();
  };
  for (int j = 0; j < nsteps; j += 4) {
    step(j, IC<0>{});
    step(j + 1, IC<1>{});
    step(j + 2, IC<2>{});
    step(j + 3, IC<3>{});
  }
  r3_wait_vm<0>();
  float* blk = a.part + ((((long long)item * vgx + bx) * vgz + bz) * 3 + wave) * (16 * 512) + lane;
#pragma unroll
  for (int t = 0; t < 16; ++t)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) blk[((t * 2 + ct) * 4 + r) * 64] = acc[t][ct][r];
}
#undef inv

// sum over items of the accumulator-order partials -> dw[co][ci][tap] (+= when accumulate): 64 partial-sum columns x 4
// item lanes per workgroup, fixed summation order.  D layout of the MFMA: lane & 15 = cin, (lane >> 4) * 4 + r = cout.
// (TrRedArgs is the public FdTrReduceJob of include/fdgan_hip.h: a launcher asked to defer fills one instead of launching)
typedef FdTrReduceJob TrRedArgs;
__device__ __forceinline__ void tr_reduce_group(const TrRedArgs& a, long long group, float (&sh)[4][64]) {
  const int col = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const long long j = group * 64 + col;   // item_stride is a multiple of 64
  const float* src = a.part + j;
  float t = 0.f;
  int s_ = ty;
  for (; s_ + 28 < a.items; s_ += 32) {
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = src[(long long)(s_ + 4 * k) * a.item_stride];
#pragma unroll
    for (int k = 0; k < 8; ++k) t += v[k];
  }
  for (; s_ < a.items; s_ += 4) t += src[(long long)s_ * a.item_stride];
  sh[ty][col] = t;
  __syncthreads();
  if (ty != 0) return;
  t = (sh[0][col] + sh[1][col]) + (sh[2][col] + sh[3][col]);
  const int lane = (int)(j & 63), r = (int)(j >> 6) & 3, ct = (int)(j >> 8) & 1;
  long long q = j >> 9;
  const int tp = (int)(q % a.taps);
  q /= a.taps;
  const int wave = (int)(q % a.nw);
  q /= a.nw;
  const int z = (int)(q % a.zt), cslice = (int)(q / a.zt);
  const int co = (z / a.kyg) * 32 + ct * 16 + (lane >> 4) * 4 + r, ci = (cslice * a.nw + wave) * 16 + (lane & 15);
  const int tap = (z % a.kyg) * a.kyn * a.ks + tp;
  if (co < a.cout && ci < a.cin) {
    float* o = a.out + ((long long)co * a.cin + ci) * (a.ks * a.ks) + tap;
    *o = a.accumulate ? *o + t : t;
  }
}
__global__ __launch_bounds__(256) void wgrad_tr_reduce_kernel(TrRedArgs a) {
  __shared__ float sh[4][64];
  tr_reduce_group(a, blockIdx.x, sh);
}
// the same sums for a whole table of jobs (a backward walk's row-walking weight gradients): one launch instead of one per conv
struct TrBatchArgs {
  const TrRedArgs* jobs;
  int njobs;
};
__global__ __launch_bounds__(256) void wgrad_tr_reduce_batch_kernel(TrBatchArgs b) {
  __shared__ float sh[4][64];
  __shared__ TrRedArgs job;
  const TrRedArgs* jobs = b.jobs;
  int lo = 0, hi = b.njobs - 1;                                 // last job whose first_group <= blockIdx.x (uniform)
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].first_group <= (long long)blockIdx.x) lo = mid;
    else hi = mid - 1;
  }
  if (threadIdx.x == 0) job = jobs[lo];
  __syncthreads();
  tr_reduce_group(job, (long long)blockIdx.x - job.first_group, sh);
}

template <int KS, int KYN, int NW, int NCO = 2>
int g3_launch(WgradRowsArgs& a, long long nimg, float* workspace, long long workspace_floats, float* dw, float* dbias, int accumulate,
              hipStream_t stream, const char* name, FdTrReduceJob* job, int defer) {
  using C = G3Cfg<KS, KYN, NW, NCO>;
  a.xblocks = (a.Wo + G3_PB - 1) / G3_PB;
  const long long strips = nimg * a.xblocks, ci_tiles = (a.Cin + NW * 16 - 1) / (NW * 16), zt = (a.Cout + 31) / 32 * (KS / KYN);
  const long long ctiles = (a.Cout + 31) / 32;
  const long long item_stride = ci_tiles * zt * NW * C::TAPS * 512, item_floats = item_stride + (dbias ? ctiles * 32 : 0);
  if (strips * item_floats > workspace_floats || strips >= 65536) return 1;   // caller falls back to the per-tap kernel
  static const char* wgs_env = FD_TUNE_GETENV("FDGAN_DEBUG_WGRAD_ITEMS");   // tuning aid
  const long long zwg = (zt + NCO / 2 - 1) / (NCO / 2);       // workgroups along z (NCO = 4: two 32-filter groups each)
  const long long base = strips * ci_tiles * zwg;
  a.zt = (int)zt;
  long long segs = (wgs_env ? atoll(wgs_env) : 256) / base;   // one resident workgroup per CU
  if (segs < 1) segs = 1;
  if (!wgs_env && a.Cout > 32 && NCO == 2) {
    // MFMA-bound shapes (D, the wide 3x3): two workgroups fit a CU (LDS), so fill 512 slots; and a grid just past a
    // multiple of the resident capacity (D's 4x4 144 -> 288: 576 workgroups = 1.125 rounds) runs a nearly empty last
    // round -- split the rows once more (measured 909 -> 805 us; 72 -> 144: 219 -> 185 us)
    if (base < 200) segs = (1024 + base - 1) / base;   // 256-workgroup grids (160 -> 128, 512 -> 128) measured WORSE when split
    else if (base % 512 != 0 && base % 512 <= 256 && base < 2048) segs = 2;
  }
  if (!wgs_env && NCO == 4) {
    // 64-filter workgroups hold ~200 registers per lane: ONE workgroup per CU (two for the 3-wave form).  Pick the row split whose
    // grid fills whole rounds of the resident capacity best (a grid just past a multiple runs a nearly empty last round); at
    // least 8 rows per item, fewer items on ties (every item writes a partial)
    const long long slots = NW <= 3 ? 512 : 256;
    double best_eff = 0.0;
    segs = 1;
    for (long long sg = 1; sg <= (a.Ho + 7) / 8; ++sg) {
      const long long wgs = base * sg;
      const double eff = (double)wgs / (double)((wgs + slots - 1) / slots * slots);
      if (eff > best_eff + 0.02) best_eff = eff, segs = sg;
    }
  }
  if (segs > (a.Ho + 1) / 2) segs = (a.Ho + 1) / 2;             // at least 2 rows per item (KYN - 1 halo rows re-staged per item)
  while (segs > 1 && (strips * segs * item_floats > workspace_floats || strips * segs >= 65536)) --segs;
  a.seg_rows = (int)((a.Ho + segs - 1) / segs);
  // The two-register-set instantiations (DEPTH 2 in the kernel: NW <= 8, NCO == 2) run their steps in PAIRS: a segment with an odd
  // row count would run one whole extra MFMA pass against a dy row loaded as zeros (+33 % matrix work for a 3-row segment) and
  // rely on 0 * x == 0 for staged inputs (an Inf / NaN activation would poison every accumulator of the item).  Even segments
  // only; the last segment of an odd Ho keeps its single tail row.
  if (NW <= 8 && NCO == 2 && (a.seg_rows & 1) && a.seg_rows < a.Ho) ++a.seg_rows;
  a.segs = (int)((a.Ho + a.seg_rows - 1) / a.seg_rows);
  a.part = workspace;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_tr_kernel<KS, KYN, NW, NCO>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) FD_FAIL(FD_ELAUNCH, "hipFuncSetAttribute(%s): %s", name, hipGetErrorString(e));
    attr_done = true;
  }
  static const char* ph = FD_TUNE_GETENV("FDGAN_DEBUG_PHASES");
  a.dbg_skip = ph ? atoi(ph) : 0;
  const long long items = strips * a.segs;
  a.bias_part = dbias ? workspace + items * item_stride : nullptr;
  if (int rc = fd_launch(&conv_wgrad_tr_kernel<KS, KYN, NW, NCO>, name, dim3((unsigned)ci_tiles, (unsigned)items, (unsigned)zwg), dim3(64 * NW),
                         C::LDS, a, stream))
    return rc;
  TrRedArgs r{workspace, dw, item_stride, (int)items, (int)zt, KS / KYN, KYN, KS, NW, C::TAPS, a.Cin, a.Cout, accumulate, 0, item_stride / 64};
  if (job) *job = r;
  if (!(job && defer && !dbias))
    if (int rc = fd_launch(&wgrad_tr_reduce_kernel, "wgrad_tr_reduce", dim3((unsigned)(item_stride / 64)), dim3(256), 0, r, stream)) return rc;
  if (!dbias) return FD_OK;
  // (one thread per channel walking all items was a chain of 300-350 dependent loads on two 64-thread workgroups: 60-80 us)
  return fd_wgrad_reduce_wide(a.bias_part, dbias, a.Cout, ctiles * 32, (int)items, accumulate, stream);
}

template <bool RELU, int DBG = 0>
int r3_launch(WgradRowsArgs& a, long long nimg, float* workspace, long long workspace_floats, float* dw, int accumulate, hipStream_t stream,
              const char* name, FdTrReduceJob* job = nullptr, int defer = 0) {
  a.xblocks = a.Wo / G3_PB;
  const long long strips = nimg * a.xblocks, ci_tiles = a.Cin / 128, zt = a.Cout / 32;
  const long long item_stride = ci_tiles * zt * 8 * 9 * 512;
  if (strips * item_stride > workspace_floats || strips >= 65536) return 1;
  static const char* wgs_env = FD_TUNE_GETENV("FDGAN_DEBUG_WGRAD_ITEMS");   // tuning aid
  const long long base = strips * ci_tiles * zt;
  // one resident workgroup per CU -- but never fewer than 8 rows per item: every item writes a 147 KB partial (and the reduction reads
  // it back) whatever it covers, so at 64 x 64 (B = 16: 1024 rows) 256 items move 37.7 MB of partials for 21 MB of data; 128 items of
  // 8 rows measured 26.4 us against 27.7 (kernel + reduction, tools/wgrad_one.py) for half of that traffic.  At 128 x 128 and above the
  // 256-item split stays faster (39.6 vs 47.6 us).
  long long want = wgs_env ? atoll(wgs_env) : fd_cus(256);
  static const char* small_env = FD_TUNE_GETENV("FDGAN_DEBUG_WGRAD_ITEMS_SMALL");   // tuning aid: the 64 x 64 item count
  if (!wgs_env && strips * ci_tiles * zt * a.Ho <= 1024) want = small_env ? atoll(small_env) : 128;
  long long segs = want / base;
  if (segs < 1) segs = 1;
  if (segs > (a.Ho + 3) / 4) segs = (a.Ho + 3) / 4;             // two of a segment's steps only see one or two of its rows
  while (segs > 1 && (strips * segs * item_stride > workspace_floats || strips * segs >= 65536)) --segs;
  a.seg_rows = (int)((a.Ho + segs - 1) / segs);
  a.segs = (int)((a.Ho + a.seg_rows - 1) / a.seg_rows);
  a.part = workspace;
  a.bias_part = nullptr;
  a.dbg_skip = 0;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_r3_kernel<RELU, DBG>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       160 * 1024);
    if (e != hipSuccess) FD_FAIL(FD_ELAUNCH, "hipFuncSetAttribute(%s): %s", name, hipGetErrorString(e));
    attr_done = true;
  }
  const long long items = strips * a.segs;
  if (int rc = fd_launch(&conv_wgrad_r3_kernel<RELU, DBG>, name, dim3((unsigned)ci_tiles, (unsigned)items, (unsigned)zt), dim3(512), R3_LDS, a, stream))
    return rc;
  TrRedArgs r{workspace, dw, item_stride, (int)items, (int)zt, 1, 3, 3, 8, 9, a.Cin, a.Cout, accumulate, 0, item_stride / 64};
  if (job) *job = r;
  if (job && defer) return FD_OK;
  return fd_launch(&wgrad_tr_reduce_kernel, "wgrad_tr_reduce", dim3((unsigned)(item_stride / 64)), dim3(256), 0, r, stream);
}

template <bool RELU>
int r4_launch(WgradRowsArgs& a, long long nimg, float* workspace, long long workspace_floats, float* dw, int accumulate, hipStream_t stream,
              const char* name, FdTrReduceJob* job = nullptr, int defer = 0) {
  a.xblocks = (a.Wo + R4_PX - 1) / R4_PX;
  const long long strips = nimg * a.xblocks, ci_tiles = a.Cin / 48, zt = a.Cout / 32;
  const long long item_stride = ci_tiles * zt * 3 * 16 * 512;
  if (strips * item_stride > workspace_floats || strips >= 65536) return 1;
  const long long base = strips * ci_tiles * zt;
  long long segs = 512 / base;                                   // two resident workgroups per CU
  if (segs < 1) segs = 1;
  if (segs > (a.Ho + 7) / 8) segs = (a.Ho + 7) / 8;              // three of a segment's steps see only part of its rows
  while (segs > 1 && (strips * segs * item_stride > workspace_floats || strips * segs >= 65536)) --segs;
  a.seg_rows = (int)((a.Ho + segs - 1) / segs);
  a.segs = (int)((a.Ho + a.seg_rows - 1) / a.seg_rows);
  a.part = workspace;
  a.bias_part = nullptr;
  a.dbg_skip = FD_TUNE_GETENV("FDGAN_DEBUG_R4_PLAIN") != nullptr ? 256 : 0;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_r4_kernel<RELU>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) FD_FAIL(FD_ELAUNCH, "hipFuncSetAttribute(%s): %s", name, hipGetErrorString(e));
    attr_done = true;
  }
  const long long items = strips * a.segs;
  a.vgx = (int)ci_tiles, a.vgy = (int)items, a.vgz = (int)zt;
  const long long grid1 = (items + 7) / 8 * 8 * ci_tiles * zt;
  if (int rc = fd_launch(&conv_wgrad_r4_kernel<RELU>, name, dim3((unsigned)grid1), dim3(192), R4_LDS, a, stream))
    return rc;
  TrRedArgs r{workspace, dw, item_stride, (int)items, (int)zt, 1, 4, 4, 3, 16, a.Cin, a.Cout, accumulate, 0, item_stride / 64};
  if (job) *job = r;
  if (job && defer) return FD_OK;
  return fd_launch(&wgrad_tr_reduce_kernel, "wgrad_tr_reduce", dim3((unsigned)(item_stride / 64)), dim3(256), 0, r, stream);
}

}  // namespace

// Which instantiation covers a stride-1 conv (0: none, the caller uses the per-tap kernel).  3x3 pad 1: 8, 5 or 3 cin
// tiles per workgroup, whichever pads Cin least (growth conv 128 -> 32, the dy blocks, the discriminator's 36 -> 72 and
// 72 -> 144); 4x4: 9 cin tiles, two filter rows per workgroup (the Fusion-discriminator's 144 -> 288 and 288 -> 1,
// /root/reference/models/dehaze1113.py:200-207).
int conv_wgrad_tr_variant(int cout, int cin, int ksize, int stride, int pad, bool pool) {
  if (stride != 1 || pool || cin < 32 || FD_TUNE_GETENV("FDGAN_DEBUG_NO_WGRAD_TR") != nullptr) return 0;
  if (ksize == 3 && pad == 1) {
    int best = 0;
    long long best_pad = 0;
    const int nws[3] = {8, 5, 3};
    for (int v = 0; v < 3; ++v) {
      const long long padded = (long long)((cin + nws[v] * 16 - 1) / (nws[v] * 16)) * nws[v] * 16;
      if (best == 0 || padded < best_pad) best = nws[v], best_pad = padded;
    }
    return best_pad * 4 <= (long long)cin * 6 ? best : 0;   // at most 1.5x padding
  }
  if (ksize == 4 && pad <= 3 && cin % 144 == 0) return 9;
  return 0;
}

/* Weight gradient into dw (+= when accumulate) through `workspace`.  Returns 1 (nothing launched) when the workspace
 * cannot hold one partial per (image, column block): the caller uses the per-tap kernel then. */
int conv_wgrad_tr_launch(int variant, WgradRowsArgs& a, long long nimg, float* workspace, long long workspace_floats, float* dw,
                         float* dbias, int accumulate, hipStream_t stream, FdTrReduceJob* job, int defer) {
  switch (variant) {
    case 8: {
      static const char* r3 = FD_TUNE_GETENV("FDGAN_DEBUG_WGRAD_R3");   // tuning aid: '0' first-generation kernel
      if (dbias == nullptr && a.Cin % 128 == 0 && a.Cout % 32 == 0 && a.Wo % G3_PB == 0 && a.W == a.Wo && !(r3 && r3[0] == '0')) {
#ifdef FDGAN_TUNING
        if (const char* dbg = FD_TUNE_GETENV("FDGAN_DEBUG_R3DBG")) {
          switch (atoi(dbg)) {
            case 1: return r3_launch<true, 1>(a, nimg, workspace, workspace_floats, dw, accumulate, stream, "conv_wgrad3x3_r3_dbg1");
            case 2: return r3_launch<true, 2>(a, nimg, workspace, workspace_floats, dw, accumulate, stream, "conv_wgrad3x3_r3_dbg2");
            case 4: return r3_launch<true, 4>(a, nimg, workspace, workspace_floats, dw, accumulate, stream, "conv_wgrad3x3_r3_dbg4");
            case 8: return r3_launch<true, 8>(a, nimg, workspace, workspace_floats, dw, accumulate, stream, "conv_wgrad3x3_r3_dbg8");
            case 6: return r3_launch<true, 6>(a, nimg, workspace, workspace_floats, dw, accumulate, stream, "conv_wgrad3x3_r3_dbg6");
            case 14: return r3_launch<true, 14>(a, nimg, workspace, workspace_floats, dw, accumulate, stream, "conv_wgrad3x3_r3_dbg14");
            case 15: return r3_launch<true, 15>(a, nimg, workspace, workspace_floats, dw, accumulate, stream, "conv_wgrad3x3_r3_dbg15");
            case 32: return r3_launch<true, 32>(a, nimg, workspace, workspace_floats, dw, accumulate, stream, "conv_wgrad3x3_r3_dbg32");
            case 64: return r3_launch<true, 64>(a, nimg, workspace, workspace_floats, dw, accumulate, stream, "conv_wgrad3x3_r3_dbg64");
            case 96: return r3_launch<true, 96>(a, nimg, workspace, workspace_floats, dw, accumulate, stream, "conv_wgrad3x3_r3_dbg96");
            case 7: return r3_launch<true, 7>(a, nimg, workspace, workspace_floats, dw, accumulate, stream, "conv_wgrad3x3_r3_dbg7");
          }
        }
#endif
        if (a.pro_mode != 0 && a.p_slope == 0.f) return r3_launch<true>(a, nimg, workspace, workspace_floats, dw, accumulate, stream, "conv_wgrad3x3_r3", job, defer);
        return r3_launch<false>(a, nimg, workspace, workspace_floats, dw, accumulate, stream, "conv_wgrad3x3_r3_leaky", job, defer);
      }
      // 96 filters and more: 64 per workgroup (round 5; measured 72 -> 144 @128^2 198 -> 152 us, 160 -> 128 224 -> 168, 640 -> 512 @32^2
      // 281 -> 219, 1024 -> 256 226 -> 160; 36 -> 72, whose second workgroup would be seven eighths padding, 75 -> 95: stays on 32);
      // FDGAN_DEBUG_WGRAD_NCO2 (tuning builds) keeps the 32-filter split
      if (a.Cout >= 96 && FD_TUNE_GETENV("FDGAN_DEBUG_WGRAD_NCO2") == nullptr)
        return g3_launch<3, 3, 8, 4>(a, nimg, workspace, workspace_floats, dw, dbias, accumulate, stream, "conv_wgrad3x3_tr8c4", job, defer);
      return g3_launch<3, 3, 8>(a, nimg, workspace, workspace_floats, dw, dbias, accumulate, stream, "conv_wgrad3x3_tr8", job, defer);
    }
    case 5:
      if (a.Cout >= 96 && FD_TUNE_GETENV("FDGAN_DEBUG_WGRAD_NCO2") == nullptr)
        return g3_launch<3, 3, 5, 4>(a, nimg, workspace, workspace_floats, dw, dbias, accumulate, stream, "conv_wgrad3x3_tr5c4", job, defer);
      return g3_launch<3, 3, 5>(a, nimg, workspace, workspace_floats, dw, dbias, accumulate, stream, "conv_wgrad3x3_tr5", job, defer);
    case 3:
      if (a.Cout >= 96 && FD_TUNE_GETENV("FDGAN_DEBUG_WGRAD_NCO2") == nullptr)
        return g3_launch<3, 3, 3, 4>(a, nimg, workspace, workspace_floats, dw, dbias, accumulate, stream, "conv_wgrad3x3_tr3c4", job, defer);
      return g3_launch<3, 3, 3>(a, nimg, workspace, workspace_floats, dw, dbias, accumulate, stream, "conv_wgrad3x3_tr3", job, defer);
    case 9: {
      static const char* r4 = FD_TUNE_GETENV("FDGAN_DEBUG_WGRAD_R4");   // tuning aid: '0' first-generation kernel
      if (dbias == nullptr && a.Cin % 48 == 0 && a.Cout % 32 == 0 && a.pad == 1 && a.Wo == a.W - 1 && a.Ho == a.H - 1 && !(r4 && r4[0] == '0')) {
        int rc;
        if (a.pro_mode != 0 && a.p_slope == 0.f) rc = r4_launch<true>(a, nimg, workspace, workspace_floats, dw, accumulate, stream, "conv_wgrad4x4_r4", job, defer);
        else rc = r4_launch<false>(a, nimg, workspace, workspace_floats, dw, accumulate, stream, "conv_wgrad4x4_r4", job, defer);
        if (rc != 1) return rc;      // 1: the workspace cannot hold this kernel's partials -- the first-generation kernel may still fit
      }
      return g3_launch<4, 2, 9>(a, nimg, workspace, workspace_floats, dw, dbias, accumulate, stream, "conv_wgrad4x4_tr", job, defer);
    }
  }
  FD_FAIL(FD_EINVAL, "conv_wgrad_tr_launch: variant %d", variant);
}

int conv_wgrad_tr_reduce_batch(const FdTrReduceJob* jobs_device, long long njobs, long long total_groups, hipStream_t stream) {
  TrBatchArgs b{jobs_device, (int)njobs};
  return fd_launch(&wgrad_tr_reduce_batch_kernel, "wgrad_tr_reduce_batch", dim3((unsigned)total_groups), dim3(256), 0, b, stream);
}
