// conv3x3_bwd.hip -- data gradient of the dense-layer growth conv (3x3, pad 1, 128 -> 32 forward, so 32 -> 128 here)
// fused with the backward of its BatchNorm + ReLU prologue, as a row-streaming kernel.
//
//   da[y][x][c] = sum_{ky,kx,k} dy[y + ky - 1][x + kx - 1][k] * Wf[ky][kx][k][c]     (Wf: the flipped filter image)
//   v = da * act'(bn(xb[y][x][c]));  sums (v, v * xb) per channel;  G = / += gamma*rstd * v   (or dpre = v)
//
// The generic implicit-GEMM instantiation moves 2.6 TB/s on this shape.  Same recipe as conv1x1_bwd.hip, organised around
// the big operands (the 128-channel xb and G rows; dy has 32 channels):
//   * a workgroup (8 waves, one per CU: the filter alone is 72 KB of LDS) owns (image, 64-pixel column block, row
//     segment) with ALL 128 output channels: the nine taps' filter fragments are staged once;
//   * dy rows live in a 4-slot LDS ring (66 pixels x 64 B): one new row per output row, requested a row ahead, ONE
//     barrier per row; a tap is a (row slot, pixel offset) pair of the same ring;
//   * wave (w & 3, w >> 2) owns 16 pixels x 64 channels: 9 B-fragment reads + 36 A-fragment reads per 36 MFMAs;
//   * row phase as in the MK kernels (accumulator tile transposed through a wave-private LDS area, whole 16-byte pieces
//     of pixel rows in and out) with the xb / G rows of the NEXT output row already requested (two register sets);
//     a lane owns the same 8 channels for the whole kernel: coefficients and BatchNorm sums in registers.
// (Round 4, measured and taken out again: the pending affine term of dy -- dy += B * y + C on the conv's own 32 output channels, 43
// fdgan_affine_accumulate launches per generator walk at 2.2 TB/s, a 64-byte slice of a 0.5-2 KB pixel pitch (tools/pitch_probe.py:
// the rate does not depend on the pitch, so it is not HBM channel camping) -- folded into the second-generation kernel's dy staging,
// the finished dy also stored dense for the weight gradient.  One more load, four coefficient reads, a stochastically rounded
// repack and a store per staged piece cost 16 spilled registers in front of the row prefetch: 209.6 us against 134.2 + 55.3 at
// 256x256, 54.9 against 37.5 + 19.8 at 128x128, 22.3 against 17.0 + 9.7 at 64x64; the step with one stream was unchanged (28.03 /
// 28.05 ms) and with the weight-gradient stream 0.9 ms LONGER, because the growth conv's weight gradient then starts behind this
// kernel instead of beside it.  Results were bitwise those of the separate pass by construction -- the parity test was not the issue.)
// Reference: autograd of conv2 / norm2 / relu2 of torchvision's _DenseLayer as used by
// /root/reference/models/dehaze1113.py:713-724.
#include <stdlib.h>

#include "conv_igemm.h"

namespace {

constexpr int B3_PB = 64;                        // output pixels per row step
constexpr int B3_DPIX = B3_PB + 2;               // staged dy pixels per row
constexpr int B3_DROW = B3_DPIX * 64;            // 32 channels x 2 B per pixel
constexpr int B3_W = 9 * 8 * 1024;               // filter fragments [tap][cout tile][lane x 16 B]
constexpr int B3_TBP = 128 + 16;                 // transposition pitch of one pixel (64 channels)
constexpr int B3_TB = 16 * B3_TBP;
constexpr int B3_LDS = B3_W + 4 * B3_DROW + 8 * B3_TB;
#define B3_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

struct Bwd3Args {
  const unsigned short* dy;   // [N][H][W][32]
  long long dy_sn;
  int dy_sh, dy_sw;
  const unsigned short* w;    // chunk32 image of the flipped filter: [tap][8 tiles][64 lanes][8]
  const unsigned short* x;    // forward input of the conv (the bottleneck), 128 channels
  long long x_sn;
  int x_sh, x_sw;
  unsigned short* g;          // gradient buffer of x, or dpre
  long long g_sn;
  int g_sh, g_sw;
  int H, W;
  int xblocks, seg_rows, segs;
  int mode, acc;              // 1 activation only, 2 BatchNorm + activation; acc 1: G += gamma*rstd*v, 2: G = gamma*rstd*v, 0: G = v
  float slope, eps;
  const float *mean, *var, *gamma, *beta;
  float* partial;             // [items][128][2] or NULL
};

__global__ __launch_bounds__(512) void conv3x3_bwd_kernel(Bwd3Args a) {
  extern __shared__ __attribute__((aligned(16))) char b3_lds[];
  char* wt = b3_lds;
  char* ring = b3_lds + B3_W;
  char* tb0 = ring + 4 * B3_DROW;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = lane & 15, kgl = lane >> 4;
  const int pxq = wave & 3, chh = wave >> 2;                  // 16-pixel quarter, 64-channel half
  char* tb = tb0 + wave * B3_TB;
  const int item = blockIdx.x;
  const int seg = item % a.segs, xb = (item / a.segs) % a.xblocks, n = item / (a.segs * a.xblocks);
  const int y_begin = seg * a.seg_rows, y_end = min(a.H, y_begin + a.seg_rows);
  const int xbase = xb * B3_PB;
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
  typedef __attribute__((ext_vector_type(4))) float f4_t;
  // ---- filter fragments: once
  for (int f = tid; f < 9 * 8 * 64; f += 512) lds_write16(wt + f * 16, *reinterpret_cast<const u32x4*>(a.w + (long long)f * 8));
  // ---- row phase ownership: 8 channels of 8 pixels per instruction, two instructions per 16-pixel tile
  const int piece = lane & 7, ql = lane >> 3;
  const int cg = chh * 64 + piece * 8;
  float sc8[8], sh8[8], s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = cg + e;
    sc8[e] = 1.f, sh8[e] = 0.f, s1[e] = s2[e] = 0.f;
    if (a.mode == 2) {
      const float gm = a.gamma ? a.gamma[c] : 1.f, bt = a.beta ? a.beta[c] : 0.f;
      sc8[e] = gm / sqrtf(a.var[c] + a.eps);
      sh8[e] = bt - a.mean[c] * sc8[e];
    }
  }
  // ---- dy row staging: thread -> (pixel tid / 4, 16-byte piece tid % 4) of the 66-pixel row
  const int dpix = tid >> 2, dpiece = tid & 3;
  const bool d_thr = dpix < B3_DPIX;
  const int dpx = xbase - 1 + dpix;
  const unsigned short* dimg = a.dy + (long long)n * a.dy_sn + dpiece * 8;
  u32x4 dyr = zero4;
  auto request_dy = [&](int row) __attribute__((always_inline)) {
    dyr = zero4;
    if (d_thr && row >= 0 && row < a.H && dpx >= 0 && dpx < a.W) dyr = *reinterpret_cast<const u32x4*>(dimg + (long long)row * a.dy_sh + (long long)dpx * a.dy_sw);
  };
  auto store_dy = [&](int row) __attribute__((always_inline)) {
    if (d_thr) lds_write16(ring + ((row + 1) & 3) * B3_DROW + dpix * 64 + dpiece * 16, dyr);
  };
  const unsigned short* ximg = a.x + (long long)n * a.x_sn + cg;
  unsigned short* gimg = a.g + (long long)n * a.g_sn + cg;
  auto request_rows = [&](int row, u32x4 (&xv)[2], u32x4 (&gv)[2]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int px = xbase + pxq * 16 + i * 8 + ql;
      xv[i] = gv[i] = zero4;
      if (row < y_end && px < a.W) {
        xv[i] = *reinterpret_cast<const u32x4*>(ximg + (long long)row * a.x_sh + (long long)px * a.x_sw);
        if (a.acc == 1) gv[i] = *reinterpret_cast<const u32x4*>(gimg + (long long)row * a.g_sh + (long long)px * a.g_sw);
      }
    }
  };
  auto step = [&](int y, u32x4 (&xv)[2], u32x4 (&gv)[2], u32x4 (&xn)[2], u32x4 (&gn)[2]) __attribute__((always_inline)) {
    store_dy(y + 1);                                          // requested during the previous step; its slot held row y - 3
    B3_BARRIER();                                             // row y + 1 visible; every wave is past the MFMAs of row y - 1
    request_dy(y + 2);
    request_rows(y + 1, xn, gn);
    // ---- MFMA: 16 pixels x 64 channels, nine taps
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const char* rowp = ring + ((y + ky) & 3) * B3_DROW;      // dy row y + ky - 1
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const bf16x8 bfr = __builtin_bit_cast(bf16x8, lds_read16(rowp + (pxq * 16 + m + kx) * 64 + kgl * 16));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bf16x8 afr = __builtin_bit_cast(bf16x8, lds_read16(wt + (((ky * 3 + kx) * 8 + chh * 4 + j) * 64 + lane) * 16));
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr, bfr, acc[j], 0, 0, 0);
        }
      }
    }
    // ---- row phase
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const u32x2 bits = __builtin_bit_cast(u32x2, __builtin_convertvector((f4_t){acc[j][0], acc[j][1], acc[j][2], acc[j][3]}, bf16x4_t));
      *reinterpret_cast<u32x2*>(tb + m * B3_TBP + j * 32 + kgl * 8) = bits;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int pl = i * 8 + ql, px = xbase + pxq * 16 + pl;
      const f32x8 da = __builtin_convertvector(__builtin_bit_cast(bf16x8, lds_read16(tb + pl * B3_TBP + piece * 16)), f32x8);
      const f32x8 fx = fd_cvt8<FmtA>(xv[i]);      // the forward input: fp16
      f32x8 o = __builtin_convertvector(__builtin_bit_cast(bf16x8, gv[i]), f32x8);
      const bool ok = px < a.W;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float pre = fmaf(fx[e], sc8[e], sh8[e]);
        const float v = ok ? da[e] * (pre > 0.f ? 1.f : a.slope) : 0.f;
        s1[e] += v;
        s2[e] += v * fx[e];
        o[e] = a.acc ? fmaf(sc8[e], v, o[e]) : v;
      }
      if (ok) *reinterpret_cast<u32x4*>(gimg + (long long)y * a.g_sh + (long long)px * a.g_sw) = __builtin_bit_cast(u32x4, __builtin_convertvector(o, bf16x8));
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };

  // rows y_begin - 1 and y_begin, then row y_begin + 1 in flight
  request_dy(y_begin - 1);
  store_dy(y_begin - 1);
  request_dy(y_begin);
  store_dy(y_begin);
  request_dy(y_begin + 1);
  u32x4 xa[2], ga[2], xb_[2], gb[2];
  request_rows(y_begin, xa, ga);
  for (int y = y_begin; y < y_end; y += 2) {
    step(y, xa, ga, xb_, gb);
    if (y + 1 < y_end) step(y + 1, xb_, gb, xa, ga);
  }
  if (a.partial != nullptr) {   // lanes 8 apart own the same channels; then the four pixel quarters in a fixed order
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
      for (int d = 8; d < 64; d <<= 1) {
        s1[e] += __shfl_xor(s1[e], d, 64);
        s2[e] += __shfl_xor(s2[e], d, 64);
      }
    B3_BARRIER();                                             // every wave is through its last row phase (red aliases tb)
    float* red = reinterpret_cast<float*>(tb0);               // [8 waves][64][2]
    if (ql == 0)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        red[(wave * 64 + piece * 8 + e) * 2] = s1[e];
        red[(wave * 64 + piece * 8 + e) * 2 + 1] = s2[e];
      }
    B3_BARRIER();
    if (tid < 256) {
      const int c = tid >> 1, which = tid & 1, h = c >> 6, cl = c & 63;   // channel half h: waves 4 h .. 4 h + 3
      const float t = (red[((4 * h + 0) * 64 + cl) * 2 + which] + red[((4 * h + 1) * 64 + cl) * 2 + which]) +
                      (red[((4 * h + 2) * 64 + cl) * 2 + which] + red[((4 * h + 3) * 64 + cl) * 2 + which]);
      a.partial[((long long)item * 128 + c) * 2 + which] = t;
    }
  }
}

// ---- second generation (image width a multiple of 64).  What the first kernel paid for: 45 LDS fragment reads per 36 MFMAs
// (the filter lived in LDS), a row phase with nothing to overlap (both waves of a SIMD reach it together), 7200 cycles per
// row.  Here:
//   * wave (w & 1, w >> 1) owns 32 pixels x 32 channels: its 18 filter fragments (9 taps x 2 channel tiles) stay in
//     REGISTERS for the whole kernel -- no filter in LDS, no A-operand reads; 18 dy-fragment reads per 36 MFMAs;
//   * the row phase of output row y - 1 (mask by act'(bn(x)), BatchNorm sums, scale, store) is issued between the MFMAs of
//     row y: the accumulator tile of a finished row waits in the wave's LDS transposition tile, so VALU and matrix pipe
//     overlap inside every wave;
//   * dy rows two steps, x rows one step ahead in registers; a 4-slot dy ring, one raw barrier per row.
// Four steps unrolled: every register set and LDS slot is a compile-time constant.
template <int V> struct IC3 { static constexpr int value = V; };
constexpr int B4_DPP = 64;                       // bytes per staged dy pixel: 32 channels x 2 B, the 16-byte piece k of pixel p stored at
                                                 // piece k ^ 2 ((p >> 2) & 1).  ds_read_b128 is serviced in four NON-contiguous 16-lane groups
                                                 // ({0-3, 12-15, 20-27}, ...: MI355X_MICROARCH.md, LDS), so a B-fragment read (lane = 16 k + m:
                                                 // pixel base + m, piece k) puts pixels m = 0-3, 12-15 of piece k and m = 4-11 of piece k + 1
                                                 // in ONE group.  With the plain 64-byte pitch AND with round 3's 80-byte pitch (which is only
                                                 // conflict-free for lanes 0-15 taken together) that is a 2-way conflict on each of the 18
                                                 // fragment reads of a row -- rocprofv3 said 1.64 / 1.73 conflict cycles per LDS cycle before
                                                 // and after that change.  tools/lds_sim.py evaluates the real groups: this XOR is the one
                                                 // (up to symmetry) that is conflict-free for every 16-pixel window at every column shift,
                                                 // and it keeps the row writes conflict-free too (the 80-byte pitch made them 2-way).
constexpr int B4_DROW = 66 * B4_DPP;             // a staged dy row: 66 pixels x 32 channels
constexpr int B4_TBP = 64;                       // transposition tile: 32 channels x 2 B per pixel, no padding: the 16-byte unit u
                                                 // of pixel p sits at u ^ ((p >> 1) & 3), which makes the ds_read_b128 lane groups
                                                 // (pixels {0,3,5,6} / {1,2,4,7} of eight) cover all 64 banks once and the
                                                 // ds_write_b64 of the accumulators 2-way at worst (pitch 80 measured 2 conflict
                                                 // cycles per LDS cycle)
constexpr int B4_TB = 32 * B4_TBP;
constexpr int B4_WF = 12 * 1024;                 // a wave's filter fragments of the second and third filter rows (round 4: the first row
                                                 // alone stays in registers -- with two rows there the kernel spilled 20-39 registers, and a
                                                 // scratch reload is a vmcnt(0) in front of the row prefetch)
constexpr int B4_LDS = 4 * B4_DROW + 8 * B4_TB + 8 * B4_WF;

// a pointer hipcc must keep in SGPRs: the loads that add a 32-bit lane offset to it take the saddr form instead of keeping one
// 64-bit lane pointer per stream alive (those were what spilled)
typedef __attribute__((address_space(1))) char* b4_gptr;   // explicitly global: an integer round trip must not turn it into a flat pointer
__device__ __forceinline__ b4_gptr b4_uniform(const unsigned short* p) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (b4_gptr)(((unsigned long long)hi << 32) | lo);
}
typedef __attribute__((address_space(1))) u32x4* b4_g16;

template <int ACC>
__global__ __launch_bounds__(512) void conv3x3_bwd2_kernel(Bwd3Args a) {
  extern __shared__ __attribute__((aligned(16))) char b3_lds[];
  char* ring = b3_lds;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, kgl = lane >> 4;
  const int pxh = wave & 1, chq = wave >> 1;                  // 32-pixel half, 32-channel quarter
  char* tb = b3_lds + 4 * B4_DROW + wave * B4_TB;
  const int item = blockIdx.x;
  const int seg = item % a.segs, xb = (item / a.segs) % a.xblocks, n = item / (a.segs * a.xblocks);
  const int y_begin = seg * a.seg_rows, y_end = min(a.H, y_begin + a.seg_rows);
  const int rows = y_end - y_begin, xbase = xb * B3_PB;
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
  typedef __attribute__((ext_vector_type(4))) float f4_t;
  // ---- filter fragments of this wave's two 16-channel tiles, once: filter row 0 stays in registers (24), rows 1 and 2 in
  // a wave-private LDS area (B4_WF)
  bf16x8 A[3][2];
  char* wf = b3_lds + 4 * B4_DROW + 8 * B4_TB + wave * B4_WF + lane * 16;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2) {
      const u32x4 f = *reinterpret_cast<const u32x4*>(a.w + ((long long)((t * 8 + chq * 2 + c2) * 64 + lane)) * 8);
      if (t < 3) A[t][c2] = __builtin_bit_cast(bf16x8, f);
      else lds_write16(wf + ((t - 3) * 2 + c2) * 1024, f);
    }
  // ---- row phase ownership: lane -> 8 channels (piece) of pixels pl0 and pl0 + 16 of the wave's 32
  const int piece = lane & 3, pl0 = lane >> 2;
  const int cg = chq * 32 + piece * 8;
  f32x2 sc2[4], sh2[4], s1[4], s2[4];                         // the lane's 8 channels as 4 pairs (fd_row8: packed fp32 math)
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = cg + e;
    float sc = 1.f, sh = 0.f;
    if (a.mode == 2) {
      const float gm = a.gamma ? a.gamma[c] : 1.f, bt = a.beta ? a.beta[c] : 0.f;
      sc = gm / sqrtf(a.var[c] + a.eps);
      sh = bt - a.mean[c] * sc;
    }
    sc2[e >> 1][e & 1] = sc, sh2[e >> 1][e & 1] = sh;
    s1[e >> 1][e & 1] = 0.f, s2[e >> 1][e & 1] = 0.f;
  }
  // ---- dy staging: thread -> (pixel tid / 4, 16-byte piece tid % 4) of the 66-pixel row; waves 0-4 (264 threads).
  // Row index rr counts from the segment's first halo row: rr = row - (y_begin - 1); slot = rr & 3.
  const int dpix = tid >> 2, dpiece = tid & 3;
  const bool d_wave = wave <= 4, d_thr = dpix < 66;
  const int dpx = xbase - 1 + dpix;
  const bool dcol = d_thr && dpx >= 0 && dpx < a.W;
  const unsigned short* dimg = a.dy + (long long)n * a.dy_sn;
  const unsigned dvo = dcol ? 2u * (unsigned)(dpx * a.dy_sw + dpiece * 8) : 0u;   // bytes
  char* dwp = ring + (d_thr ? dpix * B4_DPP + ((dpiece ^ (2 * ((dpix >> 2) & 1))) << 4) : 0);
  u32x4 dyr[2];
  unsigned dym[2];
  auto request_dy = [&](int rr, auto S) __attribute__((always_inline)) {
    constexpr int s_ = decltype(S)::value;
    // every wave, not only the five that stage the row (the others fetch the row's first 16 bytes, all lanes the same line):
    // a wave-uniform `if` around this load is control flow between a load and its use, and hipcc's vmcnt counts go
    // conservative for the whole step
    const int row = y_begin - 1 + rr;
    const bool rok = row >= 0 && row < a.H && rr <= rows + 1;
    dyr[s_] = *(b4_g16)(b4_uniform(dimg + (long long)min(max(row, 0), a.H - 1) * a.dy_sh) + dvo);
    dym[s_] = rok && dcol ? 0xffffffffu : 0u;
  };
  auto store_dy = [&](auto S, auto SLOT) __attribute__((always_inline)) {
    constexpr int s_ = decltype(S)::value, sl = decltype(SLOT)::value;
    u32x4 v = dyr[s_];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] &= dym[s_];
    if (d_thr) lds_write16(dwp + sl * B4_DROW, v);
  };
  // ---- x / G rows: two 16-byte units per lane (pixels pl0, pl0 + 16), wave-uniform row pointer + 32-bit lane offset
  const unsigned short* ximg = a.x + (long long)n * a.x_sn;
  unsigned short* gimg = a.g + (long long)n * a.g_sn;
  unsigned xvo[2], gvo[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int px = xbase + pxh * 32 + pl0 + 16 * i;
    xvo[i] = 2u * (unsigned)(px * a.x_sw + cg);   // bytes
    gvo[i] = 2u * (unsigned)(px * a.g_sw + cg);
  }
  u32x4 xs[2][2], gs[2][2];
  auto request_rows = [&](int row, auto S) __attribute__((always_inline)) {   // clamped to the segment: always readable
    constexpr int s_ = decltype(S)::value;
    const int rc = min(max(row, y_begin), y_end - 1);
    const b4_gptr xrow = b4_uniform(ximg + (long long)rc * a.x_sh);
    const b4_gptr grow = b4_uniform(gimg + (long long)rc * a.g_sh);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      xs[s_][i] = *(b4_g16)(xrow + xvo[i]);
      if constexpr (ACC == 1) gs[s_][i] = *(b4_g16)(grow + gvo[i]);
    }
  };
  // B fragment (pixel tile t, shift kx) of a staged row = 16 pixels x 64 B starting at pixel 32 pxh + 16 t + kx
  // (per column shift kx: the swizzle depends on bit 2 of the pixel index, i.e. of m + kx -- 32 pxh and 16 t do not touch it)
  const char* bptr[3];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) bptr[kx] = ring + (pxh * 32 + m + kx) * B4_DPP + ((kgl ^ (2 * (((m + kx) >> 2) & 1))) << 4);
  f32x4 acc[2][2];
#pragma unroll
  for (int t = 0; t < 2; ++t) acc[t][0] = acc[t][1] = f32x4{0.f, 0.f, 0.f, 0.f};

  // one unit (8 channels of one pixel) of the row phase of output row yr: inputs in registers, result returned
  // (w1, w0): the activation's two slopes, or (0, 0) when the step has no row to finish -- uniform VALUES, not a branch
  auto row_unit = [&](u32x4 dat, u32x4 xv, u32x4 gv, float w1, float w0) __attribute__((always_inline)) -> u32x4 {
    const f32x8 da = __builtin_convertvector(__builtin_bit_cast(bf16x8, dat), f32x8);
    const f32x8 fx = fd_cvt8<FmtA>(xv);         // the forward input: fp16
    f32x8 o, unused;
    if constexpr (ACC == 1) o = __builtin_convertvector(__builtin_bit_cast(bf16x8, gv), f32x8);
    fd_row8<ACC, false>(da, fx, o, sc2, sh2, w1, w0, s1, s2, unused);
    return __builtin_bit_cast(u32x4, __builtin_convertvector(o, bf16x8));
  };

  // ---- prologue: dy rows rr = 0 .. 2 in LDS, rr = 3, 4 requested; x (and G) of the first row requested
  request_dy(0, IC3<0>{});
  request_dy(1, IC3<1>{});
  store_dy(IC3<0>{}, IC3<0>{});
  store_dy(IC3<1>{}, IC3<1>{});
  request_dy(2, IC3<0>{});
  store_dy(IC3<0>{}, IC3<2>{});
  request_dy(3, IC3<0>{});
  request_rows(y_begin, IC3<0>{});
  request_rows(y_begin, IC3<1>{});
  for (int q = lane; q < B4_TB / 16; q += 64) lds_write16(tb + q * 16, zero4);   // read by the first step's (disabled) row phase
  B3_BARRIER();

  // step j (phase P = j mod 4): MFMAs of output row y_begin + j from the dy rows rr = j, j + 1, j + 2, interleaved with the
  // row phase of output row y_begin + j - 1
  // RP: 1 the step has a row to finish (1 <= j <= rows: the steady state, no branch around its stores), 0 it has not, 2 decide at run time
  auto step = [&](int j, auto P, auto RP) __attribute__((always_inline)) {
    constexpr int p = decltype(P)::value, p2 = p & 1, rpc = decltype(RP)::value;
    // dy: row rr = j + 3 (requested a step ago) into the slot row rr = j - 1 left; then request rr = j + 4
    store_dy(IC3<0>{}, IC3<(p + 3) & 3>{});
    request_dy(j + 4, IC3<0>{});
    const int yr = y_begin + j - 1;
    const bool rp_ok = rpc == 2 ? (j >= 1 && j <= rows) : rpc == 1;
    const float w1 = rp_ok ? 1.f : 0.f, w0 = rp_ok ? a.slope : 0.f;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      u32x4 dat = zero4;
      if (ky < 2) dat = lds_read16(tb + (pl0 + 16 * ky) * B4_TBP + ((piece ^ ((pl0 >> 1) & 3)) << 4));
      const int rowo = ((p + ky) & 3) * B4_DROW;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const bf16x8 bfr = __builtin_bit_cast(bf16x8, lds_read16(bptr[kx] + (rowo + 16 * t * B4_DPP)));
#pragma unroll
          for (int c2 = 0; c2 < 2; ++c2) {
            const bf16x8 afr = ky < 1 ? A[kx][c2] : __builtin_bit_cast(bf16x8, lds_read16(wf + (((ky - 1) * 3 + kx) * 2 + c2) * 1024));
            acc[t][c2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr, bfr, acc[t][c2], 0, 0, 0);
          }
        }
      u32x4 outv = zero4;
      if (ky < 2) outv = row_unit(dat, xs[p2 ^ 1][ky], gs[p2 ^ 1][ky], w1, w0);   // one unit of the row phase per 12-MFMA block
      // issue order of the block: two fragment reads ahead, then per fragment its two MFMAs with the row phase's VALU work
      // in their shadow and the read of the fragment after next (unconstrained, hipcc hoists all 18 reads and spills)
      // (ky = 0: filter fragments in registers; ky = 1, 2: six more fragment reads from the wave's LDS area; ky = 0, 1: a row unit)
      if (ky < 1) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
      else __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (ky < 2) __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
        else __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (ky < 2) __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
        else __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
        if (q < 4) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        if (ky >= 1 && q < 4) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (ky < 2 && rp_ok) *(b4_g16)(b4_uniform(gimg + (long long)yr * a.g_sh) + gvo[ky]) = outv;
    }
    // ---- this row's accumulators -> the transposition tile (bf16), accumulators cleared
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int c2 = 0; c2 < 2; ++c2) {
        const u32x2 bits = __builtin_bit_cast(u32x2, __builtin_convertvector((f4_t){acc[t][c2][0], acc[t][c2][1], acc[t][c2][2], acc[t][c2][3]}, bf16x4_t));
        *reinterpret_cast<u32x2*>(tb + (16 * t + m) * B4_TBP + (((c2 * 2 + (kgl >> 1)) ^ ((m >> 1) & 3)) << 4) + (kgl & 1) * 8) = bits;
        acc[t][c2] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    request_rows(y_begin + j + 1, IC3<p2 ^ 1>{});   // x of the NEXT step's output row: its row phase runs two steps from now
    B3_BARRIER();
  };
  const int nloop = (rows + 3) & ~3;
  int j = 0;
  if (rows >= 3) {
    step(0, IC3<0>{}, IC3<0>{});
    step(1, IC3<1>{}, IC3<1>{});
    step(2, IC3<2>{}, IC3<1>{});
    step(3, IC3<3>{}, IC3<1>{});
    for (j = 4; j + 3 <= rows; j += 4) {   // steady state
      step(j, IC3<0>{}, IC3<1>{});
      step(j + 1, IC3<1>{}, IC3<1>{});
      step(j + 2, IC3<2>{}, IC3<1>{});
      step(j + 3, IC3<3>{}, IC3<1>{});
    }
  }
  for (; j < nloop; j += 4) {
    step(j, IC3<0>{}, IC3<2>{});
    step(j + 1, IC3<1>{}, IC3<2>{});
    step(j + 2, IC3<2>{}, IC3<2>{});
    step(j + 3, IC3<3>{}, IC3<2>{});
  }
  if (nloop == rows) {   // the last row's phase (otherwise it ran inside one of the padding steps)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const u32x4 dat = lds_read16(tb + (pl0 + 16 * i) * B4_TBP + ((piece ^ ((pl0 >> 1) & 3)) << 4));
      const u32x4 outv = row_unit(dat, xs[1][i], gs[1][i], 1.f, a.slope);   // rows is a multiple of 4: step rows - 1 left its row in set 1... see request_rows
      *(b4_g16)(b4_uniform(gimg + (long long)(y_end - 1) * a.g_sh) + gvo[i]) = outv;
    }
  }
  if (a.partial != nullptr) {   // lanes 4 apart own the same channels; then the two pixel halves in a fixed order
    float t1[8], t2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      t1[e] = s1[e >> 1][e & 1], t2[e] = s2[e >> 1][e & 1];
#pragma unroll
      for (int d = 4; d < 64; d <<= 1) {
        t1[e] += __shfl_xor(t1[e], d, 64);
        t2[e] += __shfl_xor(t2[e], d, 64);
      }
    }
    B3_BARRIER();                                             // every wave is through its last row phase (red aliases the tiles)
    float* red = reinterpret_cast<float*>(b3_lds + 4 * B4_DROW);   // [8 waves][32][2]
    if (pl0 == 0)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        red[(wave * 32 + piece * 8 + e) * 2] = t1[e];
        red[(wave * 32 + piece * 8 + e) * 2 + 1] = t2[e];
      }
    B3_BARRIER();
    if (tid < 256) {
      const int c = tid >> 1, which = tid & 1, q = c >> 5, cl = c & 31;   // channel quarter q: waves 2 q, 2 q + 1
      const float t = red[((2 * q) * 32 + cl) * 2 + which] + red[((2 * q + 1) * 32 + cl) * 2 + which];
      a.partial[((long long)item * 128 + c) * 2 + which] = t;
    }
  }
}

}  // namespace

bool conv3x3_bwd_fits(const FdTensor* dy, const FdTensor* fwd_x, const FdTensor* dpre, const FdConvDesc* d) {
  auto rows16 = [](const FdTensor* t) {
    return t->stride[3] == 1 && t->stride[2] % 8 == 0 && t->stride[1] % 8 == 0 && t->stride[0] % 8 == 0 && ((uintptr_t)t->ptr & 15) == 0 &&
           t->stride[1] < (1ll << 31) && t->stride[2] < (1ll << 31);
  };
  return d->ksize == 3 && d->stride == 1 && d->pad == 1 && dy->c == 32 && dpre->c == 128 && fwd_x->c >= 128 && dy->h == dpre->h &&
         dy->w == dpre->w && dy->n == dpre->n && rows16(dy) && rows16(fwd_x) && rows16(dpre) && FD_TUNE_GETENV("FDGAN_DEBUG_NO_BWD3X3S") == nullptr;
}

int conv3x3_bwd_launch(const FdTensor* dy, const void* w_packed, const FdTensor* fwd_x, const FdPrologue* pro, const FdTensor* dpre,
                       int accumulate, float* partial, long long capacity_floats, long long* rows_out, long long* cpad_out,
                       hipStream_t stream) {
  Bwd3Args a{};
  a.dy = static_cast<const unsigned short*>(dy->ptr), a.dy_sn = dy->stride[0], a.dy_sh = (int)dy->stride[1], a.dy_sw = (int)dy->stride[2];
  a.w = static_cast<const unsigned short*>(w_packed);
  a.x = static_cast<const unsigned short*>(fwd_x->ptr), a.x_sn = fwd_x->stride[0], a.x_sh = (int)fwd_x->stride[1], a.x_sw = (int)fwd_x->stride[2];
  a.g = static_cast<unsigned short*>(dpre->ptr), a.g_sn = dpre->stride[0], a.g_sh = (int)dpre->stride[1], a.g_sw = (int)dpre->stride[2];
  a.H = (int)dpre->h, a.W = (int)dpre->w;
  const bool norm = pro && pro->mean;
  const int act = pro ? pro->act : FD_ACT_NONE;
  a.mode = norm ? 2 : 1, a.acc = accumulate;
  a.slope = act == FD_ACT_RELU ? 0.f : (act == FD_ACT_LEAKY02 ? 0.2f : 1.f);
  if (norm) a.mean = pro->mean, a.var = pro->var, a.gamma = pro->gamma, a.beta = pro->beta, a.eps = pro->eps;
  a.xblocks = (a.W + B3_PB - 1) / B3_PB;
  const long long strips = dpre->n * a.xblocks;
  long long segs = fd_cus(256) / strips;                       // one resident workgroup per CU (of the budget)
  if (segs < 1) segs = 1;
  if (segs > (a.H + 3) / 4) segs = (a.H + 3) / 4;              // at least 4 rows per item (2 halo rows re-staged per item)
  a.seg_rows = (int)((a.H + segs - 1) / segs);
  a.segs = (a.H + a.seg_rows - 1) / a.seg_rows;
  const long long items = strips * a.segs;
  a.partial = norm ? partial : nullptr;
  if (norm) FD_REQUIRE(partial && items * 256 <= capacity_floats, "conv2d_bwd_data: workspace too small (%lld floats needed)", items * 256);
  if (rows_out) *rows_out = items;
  if (cpad_out) *cpad_out = 128;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) FD_FAIL(FD_ELAUNCH, "hipFuncSetAttribute(conv3x3_bwd): %s", hipGetErrorString(e));
    attr_done = true;
  }
  if (a.W % B3_PB == 0 && FD_TUNE_GETENV("FDGAN_DEBUG_BWD3_V1") == nullptr) {   // second-generation kernel
    switch (a.acc) {
      case 0: return fd_launch(&conv3x3_bwd2_kernel<0>, "conv3x3_bwd_stream2", dim3((unsigned)items), dim3(512), B4_LDS, a, stream);
      case 1: return fd_launch(&conv3x3_bwd2_kernel<1>, "conv3x3_bwd_stream2", dim3((unsigned)items), dim3(512), B4_LDS, a, stream);
      default: return fd_launch(&conv3x3_bwd2_kernel<2>, "conv3x3_bwd_stream2", dim3((unsigned)items), dim3(512), B4_LDS, a, stream);
    }
  }
  return fd_launch(&conv3x3_bwd_kernel, "conv3x3_bwd_stream", dim3((unsigned)items), dim3(512), B3_LDS, a, stream);
}
