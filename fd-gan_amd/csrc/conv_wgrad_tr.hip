// conv_wgrad_tr.hip -- weight gradient of stride-1 k x k convs with gfx950's LDS transpose read: the dense-layer
// growth conv (3x3, pad 1, 32 filters, Cin a multiple of 128) and the Fusion-discriminator's 4x4 144 -> 288 conv.
//
//   dW[co][ci][ky][kx] = sum over (n, y, x) of dy[n][y][x][co] * a[n][y + ky - 1][x + kx - 1][ci],
//   a = relu(bn(x)) recomputed from the raw input exactly as the forward conv staged it (bf16, zero padding)
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
template <int KS, int KYN, int NW>
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
  static constexpr int DROW_B = G3_PB * 64;              // 32 filters x 2 B per pixel
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

template <int KS, int KYN, int NW>
__global__ __launch_bounds__(64 * NW) void conv_wgrad_tr_kernel(WgradRowsArgs a) {
  using C = G3Cfg<KS, KYN, NW>;
  extern __shared__ __attribute__((aligned(16))) char g3_lds[];
  char* Xs = g3_lds;
  char* Ds = g3_lds + C::XSLOTS * C::XROW_B;
  float* sc_s = reinterpret_cast<float*>(g3_lds + C::SC_OFF);
  float* sh_s = sc_s + NW * 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int KYG = KS / KYN;   // filter-row groups
  const int ci0 = blockIdx.x * (NW * 16), co0 = ((int)blockIdx.z / KYG) * 32, ky0 = ((int)blockIdx.z % KYG) * KYN;
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
  static_assert(C::NT >= 256 || NW == 3, "dy row staging: 256 units");
  constexpr int DK = C::NT >= 256 ? 1 : 2;        // dy units per thread (192 threads: 2 rounds, the second partly idle)
  const int dpiece = tid & 3;
  const bool dc_ok = co0 + dpiece * 8 < a.Cout;
  const unsigned short* ximg = a.x + (long long)n * a.x_sn + ci0 + xchunk * 8;
  const unsigned short* dimg = a.dy + (long long)n * a.dy_sn + co0 + dpiece * 8;
  int xdst[C::XK];
#pragma unroll
  for (int k = 0; k < C::XK; ++k) xdst[k] = C::xoff(xpix0 + XPSTEP * k, xchunk >> 1) + ((xchunk & 1) << 4);
  u32x4 xr[C::XK], dr[DK];
  // bias gradient = per-channel sum of dy: the workgroups of cin slice 0 / filter-row group 0 add up the dy rows they stage
  const bool want_bias = a.bias_part != nullptr && blockIdx.x == 0 && ky0 == 0;
  float bs[DK][8];
#pragma unroll
  for (int k = 0; k < DK; ++k)
#pragma unroll
    for (int e = 0; e < 8; ++e) bs[k][e] = 0.f;
  unsigned xok = 0;   // bit k: xr[k] holds raw data (else the unit is zero padding)
  auto load_x_row = [&](int row) __attribute__((always_inline)) {
    xok = 0;
    const bool rok = row >= 0 && row < a.H && xc_ok;
#pragma unroll
    for (int k = 0; k < C::XK; ++k) {
      const int pix = xpix0 + XPSTEP * k, px = xbase - a.pad + pix;
      xr[k] = zero4;
      if (rok && px >= 0 && px < a.W && pix < C::XPIX) {
        xr[k] = *reinterpret_cast<const u32x4*>(ximg + (long long)row * a.x_sh + (long long)px * a.x_sw);
        xok |= 1u << k;
      }
    }
  };
  auto store_x_row = [&](int row) __attribute__((always_inline)) {
    char* slot = Xs + ((unsigned)(row - row_off) % C::XSLOTS) * C::XROW_B;
#pragma unroll
    for (int k = 0; k < C::XK; ++k) {
      if (xpix0 + XPSTEP * k >= C::XPIX) break;
      u32x4 v = xr[k];
      if ((xok >> k) & 1) {
        if (a.pro_mode != 0) v = fd_xform8(v, sc_s + xchunk * 8, sh_s + xchunk * 8, a.p_slope);
      } else {
        v = zero4;   // zero padding of the ACTIVATED input
      }
      lds_write16(slot + xdst[k], v);
    }
  };
  auto load_d_row = [&](int row) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < DK; ++k) {
      const int dpix = (tid + k * C::NT) >> 2;
      dr[k] = zero4;
      if (dpix < G3_PB && dc_ok && row < y_end && xbase + dpix < a.Wo)
        dr[k] = *reinterpret_cast<const u32x4*>(dimg + (long long)row * a.dy_sh + (long long)(xbase + dpix) * a.dy_sw);
    }
  };
  auto store_d_row = [&](int row) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < DK; ++k) {
      const int dpix = (tid + k * C::NT) >> 2;
      if (dpix < G3_PB) lds_write16(Ds + (row & 1) * C::DROW_B + g3_doff(dpix, dpiece >> 1) + ((dpiece & 1) << 4), dr[k]);
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
  f32x4 acc[C::TAPS][2];
#pragma unroll
  for (int t = 0; t < C::TAPS; ++t) acc[t][0] = acc[t][1] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nsub = (min(G3_PB, a.Wo - xbase) + 31) / 32;

  // input rows of the first output row, and its dy row
  for (int r = y_begin + row_off; r < y_begin + row_off + KYN; ++r) {
    load_x_row(r);
    store_x_row(r);
  }
  load_d_row(y_begin);
  store_d_row(y_begin);
  __syncthreads();

  for (int y = y_begin; y < y_end; ++y) {
    if (!(a.dbg_skip & 4)) {
      load_x_row(y + row_off + KYN);   // in flight during this row's MFMAs
      load_d_row(y + 1);
    }
    const char* dcur = Ds + (y & 1) * C::DROW_B;
    if (!(a.dbg_skip & 2))
    for (int sub = 0; sub < nsub; ++sub) {
      const int pa = 32 * sub + kpix;
      const bf16x8 af0 = g3_frag(dcur + g3_doff(pa, 0) + piece, dcur + g3_doff(pa + 4, 0) + piece);
      const bf16x8 af1 = g3_frag(dcur + g3_doff(pa, 1) + piece, dcur + g3_doff(pa + 4, 1) + piece);
#pragma unroll
      for (int ky = 0; ky < KYN; ++ky) {
        const char* slot = Xs + ((unsigned)(y + ky) % C::XSLOTS) * C::XROW_B;   // input row y + row_off + ky
#pragma unroll
        for (int kx = 0; kx < KS; ++kx) {
          const int pb = pa + kx;   // staged pixel index = output pixel + kx (staged pixel 0 is x = xbase - pad)
          const bf16x8 bfr = g3_frag(slot + C::xoff(pb, wave) + piece, slot + C::xoff(pb + 4, wave) + piece);
          acc[ky * KS + kx][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af0, bfr, acc[ky * KS + kx][0], 0, 0, 0);
          acc[ky * KS + kx][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af1, bfr, acc[ky * KS + kx][1], 0, 0, 0);
        }
      }
    }
    if (!(a.dbg_skip & 4)) store_x_row(y + row_off + KYN);   // its slot held the row above this step's first: last read one barrier ago
    store_d_row(y + 1);
    __syncthreads();
  }
  if (want_bias) {   // after the loop's last barrier: the x slots are free
    float* red = reinterpret_cast<float*>(Xs);   // [64 pixels][32 channels]
#pragma unroll
    for (int k = 0; k < DK; ++k) {
      const int dpix = (tid + k * C::NT) >> 2;
      if (dpix < G3_PB)
#pragma unroll
        for (int e = 0; e < 8; ++e) red[dpix * 32 + dpiece * 8 + e] = bs[k][e];
    }
    __syncthreads();
    if (tid < 32) {
      float t = 0.f;
      for (int q = 0; q < G3_PB; ++q) t += red[q * 32 + tid];
      a.bias_part[((long long)item * (gridDim.z / KYG) + blockIdx.z / KYG) * 32 + tid] = t;
    }
  }
  // partial sums in accumulator order, [item][cin slice][z][wave][tap][cout tile][r][lane]: every store is 256
  // contiguous bytes (scattering them to [co][ci][tap] here cost 50 us per launch); wgrad_tr_reduce_kernel maps back
  if (!(a.dbg_skip & 1)) {
    float* blk = a.part + ((((long long)item * gridDim.x + blockIdx.x) * gridDim.z + blockIdx.z) * NW + wave) * (C::TAPS * 512) + lane;
#pragma unroll
    for (int t = 0; t < C::TAPS; ++t)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) blk[((t * 2 + ct) * 4 + r) * 64] = acc[t][ct][r];
  }
}

// sum over items of the accumulator-order partials -> dw[co][ci][tap] (+= when accumulate): 64 partial-sum columns x 4
// item lanes per workgroup, fixed summation order.  D layout of the MFMA: lane & 15 = cin, (lane >> 4) * 4 + r = cout.
struct TrRedArgs {
  const float* part;
  float* out;
  long long item_stride;   // floats per item = cin slices * z * NW * TAPS * 512
  int items, zt, kyg, kyn, ks, nw, taps, Cin, Cout, accumulate;
};
__global__ __launch_bounds__(256) void wgrad_tr_reduce_kernel(TrRedArgs a) {
  __shared__ float sh[4][64];
  const int col = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const long long j = (long long)blockIdx.x * 64 + col;   // item_stride is a multiple of 64
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
  if (co < a.Cout && ci < a.Cin) {
    float* o = a.out + ((long long)co * a.Cin + ci) * (a.ks * a.ks) + tap;
    *o = a.accumulate ? *o + t : t;
  }
}

struct TrBiasRedArgs {
  const float* part;   // [items][ctiles * 32]
  float* out;
  int items, width, Cout, accumulate;
};
__global__ __launch_bounds__(64) void wgrad_tr_bias_reduce_kernel(TrBiasRedArgs a) {
  const int co = blockIdx.x * 64 + threadIdx.x;
  if (co >= a.Cout) return;
  float t = 0.f;
  for (int s_ = 0; s_ < a.items; ++s_) t += a.part[(long long)s_ * a.width + co];
  a.out[co] = a.accumulate ? a.out[co] + t : t;
}

template <int KS, int KYN, int NW>
int g3_launch(WgradRowsArgs& a, long long nimg, float* workspace, long long workspace_floats, float* dw, float* dbias, int accumulate,
              hipStream_t stream, const char* name) {
  using C = G3Cfg<KS, KYN, NW>;
  a.xblocks = (a.Wo + G3_PB - 1) / G3_PB;
  const long long strips = nimg * a.xblocks, ci_tiles = (a.Cin + NW * 16 - 1) / (NW * 16), zt = (a.Cout + 31) / 32 * (KS / KYN);
  const long long ctiles = (a.Cout + 31) / 32;
  const long long item_stride = ci_tiles * zt * NW * C::TAPS * 512, item_floats = item_stride + (dbias ? ctiles * 32 : 0);
  if (strips * item_floats > workspace_floats || strips >= 65536) return 1;   // caller falls back to the per-tap kernel
  static const char* wgs_env = FD_TUNE_GETENV("FDGAN_DEBUG_WGRAD_ITEMS");   // tuning aid
  const long long base = strips * ci_tiles * zt;
  long long segs = (wgs_env ? atoll(wgs_env) : 256) / base;   // one resident workgroup per CU
  if (segs < 1) segs = 1;
  if (!wgs_env && a.Cout > 32) {
    // MFMA-bound shapes (D, the wide 3x3): two workgroups fit a CU (LDS), so fill 512 slots; and a grid just past a
    // multiple of the resident capacity (D's 4x4 144 -> 288: 576 workgroups = 1.125 rounds) runs a nearly empty last
    // round -- split the rows once more (measured 909 -> 805 us; 72 -> 144: 219 -> 185 us)
    if (base < 200) segs = (1024 + base - 1) / base;   // 256-workgroup grids (160 -> 128, 512 -> 128) measured WORSE when split
    else if (base % 512 != 0 && base % 512 <= 256 && base < 2048) segs = 2;
  }
  if (segs > (a.Ho + 1) / 2) segs = (a.Ho + 1) / 2;             // at least 2 rows per item (KYN - 1 halo rows re-staged per item)
  while (segs > 1 && (strips * segs * item_floats > workspace_floats || strips * segs >= 65536)) --segs;
  a.seg_rows = (int)((a.Ho + segs - 1) / segs);
  a.segs = (int)((a.Ho + a.seg_rows - 1) / a.seg_rows);
  a.part = workspace;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_tr_kernel<KS, KYN, NW>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) FD_FAIL(FD_ELAUNCH, "hipFuncSetAttribute(%s): %s", name, hipGetErrorString(e));
    attr_done = true;
  }
  static const char* ph = FD_TUNE_GETENV("FDGAN_DEBUG_PHASES");
  a.dbg_skip = ph ? atoi(ph) : 0;
  const long long items = strips * a.segs;
  a.bias_part = dbias ? workspace + items * item_stride : nullptr;
  if (int rc = fd_launch(&conv_wgrad_tr_kernel<KS, KYN, NW>, name, dim3((unsigned)ci_tiles, (unsigned)items, (unsigned)zt), dim3(64 * NW),
                         C::LDS, a, stream))
    return rc;
  TrRedArgs r{workspace, dw, item_stride, (int)items, (int)zt, KS / KYN, KYN, KS, NW, C::TAPS, a.Cin, a.Cout, accumulate};
  if (int rc = fd_launch(&wgrad_tr_reduce_kernel, "wgrad_tr_reduce", dim3((unsigned)(item_stride / 64)), dim3(256), 0, r, stream)) return rc;
  if (!dbias) return FD_OK;
  TrBiasRedArgs rb{a.bias_part, dbias, (int)items, (int)(ctiles * 32), a.Cout, accumulate};
  return fd_launch(&wgrad_tr_bias_reduce_kernel, "wgrad_tr_bias_reduce", dim3((unsigned)((a.Cout + 63) / 64)), dim3(64), 0, rb, stream);
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
                         float* dbias, int accumulate, hipStream_t stream) {
  switch (variant) {
    case 8: return g3_launch<3, 3, 8>(a, nimg, workspace, workspace_floats, dw, dbias, accumulate, stream, "conv_wgrad3x3_tr8");
    case 5: return g3_launch<3, 3, 5>(a, nimg, workspace, workspace_floats, dw, dbias, accumulate, stream, "conv_wgrad3x3_tr5");
    case 3: return g3_launch<3, 3, 3>(a, nimg, workspace, workspace_floats, dw, dbias, accumulate, stream, "conv_wgrad3x3_tr3");
    case 9: return g3_launch<4, 2, 9>(a, nimg, workspace, workspace_floats, dw, dbias, accumulate, stream, "conv_wgrad4x4_tr");
  }
  FD_FAIL(FD_EINVAL, "conv_wgrad_tr_launch: variant %d", variant);
}
