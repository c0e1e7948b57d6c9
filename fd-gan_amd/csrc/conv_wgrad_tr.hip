// conv_wgrad_tr.hip -- weight gradient of the dense-layer growth conv (3x3, stride 1, pad 1, 32 filters,
// Cin a multiple of 128) with gfx950's LDS transpose read.
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
// touches (p..p+3, p+8..p+11, any tap shift) fall in eight different 32-byte bank groups.
// Reference: autograd of the growth conv of torchvision's _DenseLayer as used by
// /root/reference/models/dehaze1113.py:713-724 (dense_block1-3).
#include <stdlib.h>

#include "conv_igemm.h"

namespace {

constexpr int G3_PB = 64;                       // output pixels per row step
constexpr int G3_XPIX = G3_PB + 2;              // staged input pixels per row (one halo pixel each side)
constexpr int G3_XROW_B = G3_XPIX * 256;        // 128 channels x 2 B per pixel
constexpr int G3_XSLOTS = 4;                    // rows y-1, y, y+1 in use + row y+2 being written
constexpr int G3_DROW_B = G3_PB * 64;           // 32 filters x 2 B per pixel
constexpr int G3_SC_OFF = G3_XSLOTS * G3_XROW_B + 2 * G3_DROW_B;
constexpr int G3_LDS = G3_SC_OFF + 2 * 128 * 4;

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ bf16x8 g3_frag(const char* p0, const char* p1) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p1));
  return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}
__device__ __forceinline__ int g3_xoff(int pix, int c16) {   // byte offset of 16-channel group c16 of staged pixel pix
  return pix * 256 + ((c16 ^ ((pix & 3) | (((pix >> 3) & 1) << 2))) << 5);
}
__device__ __forceinline__ int g3_doff(int pix, int c16) {
  return pix * 64 + ((c16 ^ ((pix >> 3) & 1)) << 5);
}

__global__ __launch_bounds__(512) void conv_wgrad3x3_tr_kernel(WgradRowsArgs a) {
  extern __shared__ __attribute__((aligned(16))) char g3_lds[];
  char* Xs = g3_lds;
  char* Ds = g3_lds + G3_XSLOTS * G3_XROW_B;
  float* sc_s = reinterpret_cast<float*>(g3_lds + G3_SC_OFF);
  float* sh_s = sc_s + 128;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ci0 = blockIdx.x * 128;
  const int item = blockIdx.y;                       // (image, column block, row segment)
  const int seg = item % a.segs, xb = (item / a.segs) % a.xblocks, n = item / (a.segs * a.xblocks);
  const int y_begin = seg * a.seg_rows, y_end = min(a.H, y_begin + a.seg_rows);
  const int xbase = xb * G3_PB;
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
  const u32x4 zero4 = {0u, 0u, 0u, 0u};

  // ---- staging map.  x: unit (pix, chunk) = (tid / 16 + 32 k, tid % 16), k = 0, 1 and k = 2 for the two halo pixels
  // 64, 65: 16 lanes cover one pixel's 256 bytes.  dy: threads 0-255, unit (pix, piece) = (tid / 4, tid % 4).
  const int xchunk = tid & 15, xpix0 = tid >> 4;
  const bool x_third = xpix0 < 2;
  const int dpix = (tid >> 2) & 63, dpiece = tid & 3;
  const bool is_d = tid < 256;
  const unsigned short* ximg = a.x + (long long)n * a.x_sn + ci0 + xchunk * 8;
  const unsigned short* dimg = a.dy + (long long)n * a.dy_sn + dpiece * 8;
  int xdst[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) xdst[k] = g3_xoff(xpix0 + 32 * k, xchunk >> 1) + ((xchunk & 1) << 4);
  const int ddst = g3_doff(dpix, dpiece >> 1) + ((dpiece & 1) << 4);

  u32x4 xr[3], dr;
  unsigned xok = 0;   // bit k: xr[k] holds raw data (else the unit is zero padding)
  auto load_x_row = [&](int row) __attribute__((always_inline)) {
    xok = 0;
    const bool rok = row >= 0 && row < a.H;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int px = xbase - 1 + xpix0 + 32 * k;
      xr[k] = zero4;
      if (rok && px >= 0 && px < a.W && (k < 2 || x_third)) {
        xr[k] = *reinterpret_cast<const u32x4*>(ximg + (long long)row * a.x_sh + (long long)px * a.x_sw);
        xok |= 1u << k;
      }
    }
  };
  auto store_x_row = [&](int row) __attribute__((always_inline)) {
    char* slot = Xs + ((row + 4) & 3) * G3_XROW_B;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      if (k == 2 && !x_third) break;
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
    dr = zero4;
    if (is_d && row < y_end && xbase + dpix < a.W)
      dr = *reinterpret_cast<const u32x4*>(dimg + (long long)row * a.dy_sh + (long long)(xbase + dpix) * a.dy_sw);
  };
  auto store_d_row = [&](int row) __attribute__((always_inline)) {
    if (is_d) lds_write16(Ds + (row & 1) * G3_DROW_B + ddst, dr);
  };

  // ---- fragment addresses: lane (g, i): k rows 8 g + (i >> 2) (+ 4 for the second read), 4-channel piece i & 3
  const int g = lane >> 4, i = lane & 15;
  const int kpix = 8 * g + (i >> 2), piece = (i & 3) * 8;
  f32x4 acc[9][2];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t][0] = acc[t][1] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nsub = (min(G3_PB, a.W - xbase) + 31) / 32;

  // rows y_begin-1 .. y_begin+1 and dy row y_begin
  for (int r = y_begin - 1; r <= y_begin + 1; ++r) {
    load_x_row(r);
    store_x_row(r);
  }
  load_d_row(y_begin);
  store_d_row(y_begin);
  __syncthreads();

  for (int y = y_begin; y < y_end; ++y) {
    load_x_row(y + 2);   // in flight during this row's MFMAs
    load_d_row(y + 1);
    const char* dcur = Ds + (y & 1) * G3_DROW_B;
    for (int sub = 0; sub < nsub; ++sub) {
      const int pa = 32 * sub + kpix;
      const bf16x8 af0 = g3_frag(dcur + g3_doff(pa, 0) + piece, dcur + g3_doff(pa + 4, 0) + piece);
      const bf16x8 af1 = g3_frag(dcur + g3_doff(pa, 1) + piece, dcur + g3_doff(pa + 4, 1) + piece);
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const char* slot = Xs + ((y + ky + 3) & 3) * G3_XROW_B;   // input row y + ky - 1
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int pb = pa + kx;   // staged pixel index = output pixel + kx (halo pixel 0 is x = xbase - 1)
          const bf16x8 bfr = g3_frag(slot + g3_xoff(pb, wave) + piece, slot + g3_xoff(pb + 4, wave) + piece);
          acc[ky * 3 + kx][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af0, bfr, acc[ky * 3 + kx][0], 0, 0, 0);
          acc[ky * 3 + kx][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af1, bfr, acc[ky * 3 + kx][1], 0, 0, 0);
        }
      }
    }
    store_x_row(y + 2);   // slot (y + 2) & 3 == (y - 2) & 3: last read one barrier ago
    store_d_row(y + 1);
    __syncthreads();
  }
  // D layout: column (lane & 15) = cin, rows (lane >> 4) * 4 + r = cout
  float* dwp = a.part + (long long)item * 32 * a.Cin * 9;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = ct * 16 + g * 4 + r, ci = ci0 + wave * 16 + i;
        dwp[((long long)co * a.Cin + ci) * 9 + t] = acc[t][ct][r];
      }
}

}  // namespace

bool conv_wgrad3x3_tr_fits(int cout, int cin, int ksize, int stride, int pad, bool pool) {
  return ksize == 3 && stride == 1 && pad == 1 && !pool && cout == 32 && cin % 128 == 0 && getenv("FDGAN_DEBUG_NO_WGRAD_TR") == nullptr;
}

/* Fills a.xblocks / segs / seg_rows and launches; a.part must hold items * 32 * Cin * 9 floats (items returned). */
int conv_wgrad3x3_tr_launch(WgradRowsArgs& a, long long nimg, long long workspace_floats, long long* items_out, hipStream_t stream) {
  a.xblocks = (a.W + G3_PB - 1) / G3_PB;
  const long long strips = nimg * a.xblocks, ci_tiles = a.Cin / 128;
  const long long numel = 32LL * a.Cin * 9;
  long long segs = 512 / (strips * ci_tiles);                   // ~2 workgroups per CU in all
  if (segs < 1) segs = 1;
  if (segs > (a.H + 1) / 2) segs = (a.H + 1) / 2;               // at least 2 rows per item (2 halo rows re-staged per item)
  while (segs > 1 && strips * segs * numel > workspace_floats) --segs;
  FD_REQUIRE(strips * segs * numel <= workspace_floats && strips * segs < 65536, "conv2d_bwd_weight: workspace too small (%lld floats)",
             strips * segs * numel);
  a.seg_rows = (int)((a.H + segs - 1) / segs);
  a.segs = (int)((a.H + a.seg_rows - 1) / a.seg_rows);
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad3x3_tr_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) FD_FAIL(FD_ELAUNCH, "hipFuncSetAttribute(conv_wgrad3x3_tr): %s", hipGetErrorString(e));
    attr_done = true;
  }
  *items_out = strips * a.segs;
  return fd_launch(&conv_wgrad3x3_tr_kernel, "conv_wgrad3x3_tr", dim3((unsigned)ci_tiles, (unsigned)(strips * a.segs)), dim3(512),
                   G3_LDS, a, stream);
}
