// conv_c1.hip -- the ONE-filter convolution that ends both discriminators (nn.Conv2d(8 nf, 1, 4, 1, 1, bias=False) behind
// LeakyReLU(0.2) and in front of the sigmoid: /root/reference/models/dehaze1113.py:221-223, models/dehaze22.py:147-149) and
// its data gradient.
//
// On the implicit-GEMM kernels a single output channel is a 16- or 32-wide MFMA tile with one live row (forward:
// conv4x4_bn32, 147 us at B = 16 @ 127x127, 31/32 of the matrix work on padding), and its data gradient is a K = 16 taps x 1
// channel problem staged as sixteen 32-channel chunks (conv4x4_bn128_bwd: 297 us for 148 MB of output).  Both are really
// vector-ALU / HBM problems:
//   forward   y[p] = sum_tap sum_c a[p + tap][c] w[tap][c]: one thread per output pixel of an 8 x 32 tile; the activated halo
//             tile of a 32-channel chunk sits in LDS as four planes of [pixel][8 channels] (consecutive lanes read consecutive
//             16-byte slots: conflict-free), the chunk's 16 x 32 weights beside it (broadcast reads); v_dot2_f32_f16, fp32 sums.
//   dgrad     dx[p][c] = act'(x[p][c]) * sum_tap dy[p + pad - tap] w[c][tap]: one thread per (pixel, 8 channels), the 16 x C
//             filter as fp32 in LDS, the mask and the gradient buffer as whole 16-byte pieces of pixel rows.
#include "conv_igemm.h"

namespace {

constexpr int C1_TH = 8, C1_TW = 32;           // output pixels per workgroup (256 threads: one each)

struct C1FwdArgs {
  const unsigned short* x;     // NHWC fp16 view
  long long x_sn;
  int x_sh, x_sw, H, W, Cin, nchunk;
  const unsigned short* w;     // chunk32 fp16 image of the [1][Cin][KS][KS] filter
  int pro_mode;
  float slope, eps;
  const float *mean, *var, *gamma, *beta;
  float* y;                    // [N][1][Ho][Wo] fp32 (pre-sigmoid)
  long long y_sn;
  int y_sh, Ho, Wo, pad, tiles_x, tiles_y;
  float e_slope;
};

template <int KS>
__global__ __launch_bounds__(256) void conv_cout1_kernel(C1FwdArgs a) {
  constexpr int IH = C1_TH + KS - 1, IW = C1_TW + KS - 1, NPIX = IH * IW, NPIXR = (NPIX + 15) / 16 * 16;
  constexpr int PLANE = NPIXR * 16, IN_B = 4 * PLANE, W_B = KS * KS * 64;
  extern __shared__ __attribute__((aligned(16))) char c1_lds[];
  char* in_lds = c1_lds;                                   // [2][4 planes][NPIXR][16 B]
  char* w_lds = c1_lds + 2 * IN_B;                          // [2][taps][4 groups][16 B]
  float* sc_lds = reinterpret_cast<float*>(w_lds + 2 * W_B);   // [nchunk * 32] scale, shift
  float* sh_lds = sc_lds + a.nchunk * 32;
  const int tid = threadIdx.x;
  int tile = blockIdx.x;
  const int tx = tile % a.tiles_x;
  tile /= a.tiles_x;
  const int ty = tile % a.tiles_y, n = tile / a.tiles_y;
  const int oy0 = ty * C1_TH, ox0 = tx * C1_TW;
  for (int c = tid; c < a.nchunk * 32; c += 256) {
    float sc = 1.f, sh = 0.f;
    if (a.pro_mode == 2) {
      sc = 0.f;
      if (c < a.Cin) {
        const float g = a.gamma ? a.gamma[c] : 1.f, b = a.beta ? a.beta[c] : 0.f;
        sc = g / sqrtf(a.var[c] + a.eps);
        sh = b - a.mean[c] * sc;
      }
    }
    sc_lds[c] = sc, sh_lds[c] = sh;
  }
  __syncthreads();
  const unsigned short* xn = a.x + (long long)n * a.x_sn;
  constexpr int UNITS = NPIX * 4, UPT = (UNITS + 255) / 256;
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  u32x4 rin[UPT], rwt;
  const int wtap = tid < KS * KS * 4 ? tid >> 2 : 0;      // the chunk's weights: tap tid / 4, 8-channel group tid % 4 (threads past them: tap 0 again)
  auto load = [&](int chunk) {
    rwt = *reinterpret_cast<const u32x4*>(a.w + ((long long)(chunk * KS * KS + wtap) * 512 + (tid & 3) * 16 * 8));   // fragment image: cout row 0 = lanes 0, 16, 32, 48
#pragma unroll
    for (int i = 0; i < UPT; ++i) {
      const int u = tid + i * 256, p = u >> 2, kg = u & 3;           // the four 16-byte groups of a pixel are adjacent lanes: 64 B runs
      const int py = p / IW, px = p - py * IW;
      const int gy = oy0 - a.pad + py, gx = ox0 - a.pad + px;
      const bool ok = u < UNITS && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W && chunk * 32 + kg * 8 < a.Cin;
      // unconditional load of a clamped address, zeroed afterwards: a load inside `ok ? load : zero` is a branch, and hipcc waits for
      // every load at the join behind it (s_waitcnt vmcnt(0)) -- the units of a chunk were fetched one after the other
      const u32x4 v = *reinterpret_cast<const u32x4*>(xn + (ok ? (long long)gy * a.x_sh + (long long)gx * a.x_sw + chunk * 32 + kg * 8 : 0));
      rin[i] = ok ? v : zero4;
    }
  };
  auto store = [&](char* buf, char* wbuf, int chunk) {
#pragma unroll
    for (int i = 0; i < UPT; ++i) {
      const int u = tid + i * 256, p = u >> 2, kg = u & 3;
      if (u >= UNITS) continue;
      const int py = p / IW, px = p - py * IW;
      const int gy = oy0 - a.pad + py, gx = ox0 - a.pad + px;
      const bool ok = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W && chunk * 32 + kg * 8 < a.Cin;
      u32x4 v = rin[i];
      if (a.pro_mode != 0) v = fd_xform8(v, sc_lds + chunk * 32 + kg * 8, sh_lds + chunk * 32 + kg * 8, a.slope);
      lds_write16(buf + kg * PLANE + p * 16, ok ? v : zero4);          // zero padding applies to the ACTIVATED input
    }
    if (tid < KS * KS * 4) lds_write16(wbuf + tid * 16, rwt);
  };
  const int ly = tid / C1_TW, lx = tid % C1_TW;
  float acc4[4] = {0.f, 0.f, 0.f, 0.f};                    // four independent chains (v_dot2c accumulates in place)
  load(0);
  store(in_lds, w_lds, 0);
  __syncthreads();
  for (int chunk = 0; chunk < a.nchunk; ++chunk) {
    const bool more = chunk + 1 < a.nchunk;
    load(more ? chunk + 1 : chunk);      // (unconditional: behind `if (more)` the loads were waited for at the join, before the taps)
    const char* ib = in_lds + (chunk & 1) * IN_B;
    const char* wb = w_lds + (chunk & 1) * W_B;
#pragma unroll 1
    for (int dy = 0; dy < KS; ++dy)        // one filter row at a time: fully unrolled, hipcc hoists all 128 fragment reads (448 B of scratch)
#pragma unroll
      for (int dx = 0; dx < KS; ++dx) {
        const int p = (ly + dy) * IW + lx + dx;
#pragma unroll
        for (int kg = 0; kg < 4; ++kg) {
          // (a bit_cast of av[q] / wv[q] straight into the builtin made hipcc 7.2 use dword 0 for every q -- four identical
          // v_dot2c_f32_f16 -- so the fragments are taken apart through a typed vector first)
          const f16x8 av = __builtin_bit_cast(f16x8, lds_read16(ib + kg * PLANE + p * 16));
          const f16x8 wv = __builtin_bit_cast(f16x8, lds_read16(wb + ((dy * KS + dx) * 4 + kg) * 16));
          acc4[0] = __builtin_amdgcn_fdot2(__builtin_shufflevector(av, av, 0, 1), __builtin_shufflevector(wv, wv, 0, 1), acc4[0], false);
          acc4[1] = __builtin_amdgcn_fdot2(__builtin_shufflevector(av, av, 2, 3), __builtin_shufflevector(wv, wv, 2, 3), acc4[1], false);
          acc4[2] = __builtin_amdgcn_fdot2(__builtin_shufflevector(av, av, 4, 5), __builtin_shufflevector(wv, wv, 4, 5), acc4[2], false);
          acc4[3] = __builtin_amdgcn_fdot2(__builtin_shufflevector(av, av, 6, 7), __builtin_shufflevector(wv, wv, 6, 7), acc4[3], false);
        }
      }
    if (more) store(in_lds + ((chunk + 1) & 1) * IN_B, w_lds + ((chunk + 1) & 1) * W_B, chunk + 1);
    __syncthreads();
  }
  const float acc = (acc4[0] + acc4[1]) + (acc4[2] + acc4[3]);
  const int oy = oy0 + ly, ox = ox0 + lx;
  if (oy < a.Ho && ox < a.Wo) a.y[(long long)n * a.y_sn + (long long)oy * a.y_sh + ox] = fmaxf(acc, a.e_slope * acc);
}

// ---- round 5: the same one-filter convolution on the matrix pipe, "taps as output channels" -------------------------------------
// conv_cout1_kernel above is bound by LDS reads (every output pixel re-reads its KS^2 x C window: 9 KB of ds_read_b128 per
// pixel, 123 us for the 148 MB of D's 127 x 127 x 288 input at B = 16).  Read the sum the other way round:
//     s[q][tap] = sum_c a[q][c] * w[tap][c]          for every INPUT pixel q: a 1x1 convolution with KS^2 <= 16 "output channels"
//     y[p]      = sum_(ky,kx) s[p + (ky,kx) - pad][ky KS + kx]                       a gather of KS^2 values per output pixel
// The first line is one v_mfma_f32_16x16x32_f16 per 16 pixels x 32 channels with the (zero-extended) taps as the 16 rows of the A
// operand -- no halo in the heavy part, every activation multiplied exactly once -- and its B fragments come straight from HBM in
// conv1x1_xs's "x-stream" form: lane (m, g) reads 32 contiguous bytes (channels 64 ks + 16 g .. + 15) of pixel m, so four lanes
// consume one 128-byte line (the filter is laid out for that k order once per workgroup).  The second line reads fp32 sums from
// LDS ([row][tap][column]: consecutive lanes, consecutive words).  Workgroup = (image, band of C1M_R output rows, block of <= 128
// input columns); its R + KS - 1 input rows are 8 x (R + KS - 1) units of 16 pixels, spread over 8 waves, each unit's 2 nks loads
// requested one unit ahead.  Only the band's KS - 1 halo rows are read twice (1.375x at R = 8; the old tile re-read 1.5x).
constexpr int C1M_R = 8, C1M_CW = 128;
template <int KS, int NKS>
__global__ __launch_bounds__(512) void conv_cout1_mfma_kernel(C1FwdArgs a) {
  constexpr int KK = KS * KS, IR = C1M_R + KS - 1, CSTEP = C1M_CW - (KS - 1), NUNIT = IR * (C1M_CW / 16), UPW = (NUNIT + 7) / 8;
  extern __shared__ __attribute__((aligned(16))) char c1_lds[];
  float* s_lds = reinterpret_cast<float*>(c1_lds);                       // [IR][16][CW] fp32
  char* w_lds = c1_lds + IR * 16 * C1M_CW * 4;                           // [NKS][2][64 lanes][16 B]: A fragments, x-stream k order
  float* sc_lds = reinterpret_cast<float*>(w_lds + NKS * 2 * 1024);      // [NKS * 64] scale, shift
  float* sh_lds = sc_lds + NKS * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, g = lane >> 4;
  int blk = blockIdx.x;
  const int cb = blk % a.tiles_x;
  blk /= a.tiles_x;
  const int band = blk % a.tiles_y, n = blk / a.tiles_y;
  const int oy0 = band * C1M_R, cx0 = cb * CSTEP;      // first output row; first input column of the block
  // ---- once per workgroup: per-channel scale / shift, the filter as A fragments (taps >= KS^2 and channels >= Cin: zero)
  for (int c = tid; c < NKS * 64; c += 512) {
    float sc = 1.f, sh = 0.f;
    if (a.pro_mode == 2) {
      sc = 0.f;
      if (c < a.Cin) {
        const float gm = a.gamma ? a.gamma[c] : 1.f, bt = a.beta ? a.beta[c] : 0.f;
        sc = gm / sqrtf(a.var[c] + a.eps);
        sh = bt - a.mean[c] * sc;
      }
    }
    sc_lds[c] = sc, sh_lds[c] = sh;
  }
  for (int i = tid; i < NKS * 2 * 64 * 8; i += 512) {
    const int e = i & 7, l = (i >> 3) & 63, j = (i >> 9) & 1, ks = i >> 10;
    const int tap = l & 15, c = ks * 64 + (l >> 4) * 16 + j * 8 + e;
    const bool ok = tap < KK && c < a.Cin;
    // packed chunk32 image of the [1][Cin][KS][KS] filter: (chunk, tap) fragments of 512 elements, cout row 0 = lanes 0, 16, 32, 48
    const unsigned short v = a.w[ok ? ((long long)((c >> 5) * KK + tap) * 512 + (((c >> 3) & 3) * 16) * 8 + (c & 7)) : 0];
    reinterpret_cast<unsigned short*>(w_lds)[i] = ok ? v : (unsigned short)0;
  }
  __syncthreads();
  constexpr bool WREG = NKS <= 5;      // the A fragments live in registers (40) or are re-read from LDS per unit (NKS = 8: 64 would spill)
  u32x4 wf[WREG ? NKS : 1][2];
  if constexpr (WREG) {
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
      for (int j = 0; j < 2; ++j) wf[ks][j] = lds_read16(w_lds + ((ks * 2 + j) * 64 + lane) * 16);
  }
  const int cmax = ((a.Cin + 7) / 8) * 8;
  const unsigned short* xn = a.x + (long long)n * a.x_sn;
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  // unit u = wave + 8 k: input row ir = u / 8 of the band, 16-pixel tile u % 8; this lane's pixel (gy, gx)
  u32x4 cur[NKS][2], nxt[NKS][2];
  auto unit_off = [&](int u, bool& inb) -> long long {
    const int ir = u >> 3, t = u & 7;
    const int gy = oy0 - a.pad + ir, gx = cx0 + t * 16 + m;
    inb = u < NUNIT && gy >= 0 && gy < a.H && gx < a.W;
    return inb ? (long long)gy * a.x_sh + (long long)gx * a.x_sw : 0;
  };
  auto fetch = [&](u32x4 (&dst)[NKS][2], long long off) {
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int c = ks * 64 + g * 16 + j * 8;
        dst[ks][j] = *reinterpret_cast<const u32x4*>(xn + off + (c < cmax ? c : 0));      // past Cin: chunk 0 (its weights are zero)
      }
  };
  bool inb_c, inb_n;
  long long off_c = unit_off(wave, inb_c);
  fetch(cur, off_c);
#pragma unroll 1
  for (int k = 0; k < UPW; ++k) {
    const int u = wave + 8 * k;
    const long long off_n = unit_off(u + 8, inb_n);
    fetch(nxt, off_n);                                  // (unconditional: a clamped address, masked below)
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int c = ks * 64 + g * 16 + j * 8;
        u32x4 v = cur[ks][j];
        if (a.pro_mode != 0) v = fd_xform8(v, sc_lds + c, sh_lds + c, a.slope);
        v = (inb_c && c < cmax) ? v : zero4;            // zero padding is post-activation; pixels outside the image contribute nothing
        const u32x4 wfr = WREG ? wf[WREG ? ks : 0][j] : lds_read16(w_lds + ((ks * 2 + j) * 64 + lane) * 16);
        acc = fd_mfma<FmtA>(wfr, v, acc);
      }
    if (u < NUNIT) {
      const int ir = u >> 3, t = u & 7;
#pragma unroll
      for (int r = 0; r < 4; ++r) s_lds[(ir * 16 + g * 4 + r) * C1M_CW + t * 16 + m] = acc[r];
    }
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
      for (int j = 0; j < 2; ++j) cur[ks][j] = nxt[ks][j];
    inb_c = inb_n;
  }
  __syncthreads();
  // ---- the gather: this block owns the outputs whose window starts at input column ws = ox - pad in [cx0, cx0 + CSTEP) (block 0: from -pad)
  const int ox_lo = cb == 0 ? 0 : cx0 + a.pad, ox_hi = min(a.Wo, cx0 + CSTEP + a.pad);
  const int ncol = ox_hi - ox_lo, nrow = min(C1M_R, a.Ho - oy0);
  for (int i = tid; i < nrow * ncol; i += 512) {
    const int ly = i / ncol, ox = ox_lo + (i - ly * ncol);
    const int wsl = ox - a.pad - cx0;                  // window start, local column (-pad .. CSTEP - 1)
    float acc = 0.f;
#pragma unroll
    for (int ky = 0; ky < KS; ++ky)
#pragma unroll
      for (int kx = 0; kx < KS; ++kx) {
        const int lc = wsl + kx;
        const float v = s_lds[((ly + ky) * 16 + ky * KS + kx) * C1M_CW + (lc >= 0 && lc < C1M_CW ? lc : 0)];
        acc += (lc >= 0 && lc < C1M_CW) ? v : 0.f;       // columns left of the image (block 0) and right of the block's 128 (image edge) are padding
      }
    a.y[(long long)n * a.y_sn + (long long)(oy0 + ly) * a.y_sh + ox] = fmaxf(acc, a.e_slope * acc);
  }
}

struct C1BwdArgs {
  const unsigned short* dy;    // NHWC bf16 view, channel 0 used
  long long dy_sn;
  int dy_sh, dy_sw, Ho, Wo;
  const unsigned short* w;     // chunk32 bf16 image of the flipped filter: cout' = C, cin' = 1
  const unsigned short* x;     // forward input (fp16), the activation mask
  long long x_sn;
  int x_sh, x_sw;
  unsigned short* g;           // gradient of x (bf16)
  long long g_sn;
  int g_sh, g_sw;
  int H, W, C, C8, ks, pad, acc;   // pad: the FORWARD conv's
  float slope;
  long long total;
  int ntile;
};

// KS x KS taps; one thread = TWO horizontally adjacent pixels x four 8-channel pieces C32 apart (piece q of lane g is g + C32 q: for a
// fixed q adjacent lanes touch adjacent 16 bytes, so every load / store instruction covers whole runs of a pixel row).
// Round 3 (PMC: 1650 vector instructions per wave and unit, half of the LDS cycles bank conflicts): the dy taps of both pixels are
// fetched first, with 32-bit offsets and zeros outside the image, so the tap loop has no branches; the FMAs are packed
// (v_pk_fma_f32, the tap value broadcast); a filter fragment read from LDS serves both pixels; and the LDS image is
// [tap][low / high half][piece][4 floats], so the lanes of a read touch consecutive 16-byte slots (the [tap][channel] image
// made every ds_read_b128 a two-way conflict).
// (Also measured in round 3: the same gradient on the matrix pipe -- K = 16 taps is one v_mfma_f32_16x16x16_bf16 per 16 pixels x 16
// channels, the filter in registers, two row-permuted tiles giving every lane eight adjacent channels: identical results, 168 us
// against this kernel's 128.  The MFMA result layout leaves a wave instruction with 16 pixels x 64 bytes, half a cache line per
// pixel, and that access pattern costs more than the arithmetic it saves; a version that transposes through LDS to whole rows is
// what it would take.)
template <int KS, bool ACC1>      // ACC1: a.acc == 1 (G += ...), its read of G requested with the mask operand, ahead of the taps
__global__ __launch_bounds__(256) void dgrad_cout1_kernel(C1BwdArgs a) {
  // Third form (round 3).  Unit = one 8-channel piece of FOUR vertically adjacent pixels; units are numbered piece-fastest, then
  // pixel, so the 64 lanes of a wave instruction touch 1 KB of ONE contiguous run of the NHWC row (whole 128-byte lines -- the
  // "pieces C32 apart" numbering of the second form gave 144-byte runs, the matrix-pipe form 64-byte ones, and on this memory
  // system that is what decides: 177 / 129 / 168 us).  The four pixels share their dy window ((4 + KS - 1) x KS values instead
  // of 4 KS^2) and every filter fragment read from LDS (2 ds_read_b128 per tap for 64 packed FMAs).
  extern __shared__ __attribute__((aligned(16))) char c1_lds[];
  float* wl = reinterpret_cast<float*>(c1_lds);            // [taps][2][C8][4] fp32, forward tap order
  constexpr int KK = KS * KS, R = 4;
  const int cp = a.C8 * 8;
  // (four independent, unconditional loads per pass: as `if (c < C) v = load` in a rolled loop this was 18 dependent loads, each
  // waited for, in front of every workgroup's ~4 units of work)
  for (int i0 = threadIdx.x; i0 < KK * cp; i0 += 4 * 256) {
    float v[4];
    int dst[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = i0 + 256 * q, ic = i < KK * cp ? i : 0;
      const int t = ic / cp, c = ic - t * cp;
      const bool ok = i < KK * cp && c < a.C;
      // flipped image: Wf[cout' = c][cin' = 0][tap'] = W[0][c][KK - 1 - tap'];  fragment order, lane = c & 15 (cin' group 0), e = 0
      const int tp = KK - 1 - t;
      const float w = fd_cvt1<FmtG>(a.w[ok ? ((long long)tp * a.ntile + (c >> 4)) * 512 + (c & 15) * 8 : 0]);
      v[q] = ok ? w : 0.f;
      dst[q] = i < KK * cp ? ((t * 2 + ((c >> 2) & 1)) * a.C8 + (c >> 3)) * 4 + (c & 3) : -1;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (dst[q] >= 0) wl[dst[q]] = v[q];
  }
  __syncthreads();
  const unsigned units = (unsigned)a.total;                 // N * ceil(H / 4) * W * C8 (launcher: < 2^31)
  const unsigned H4 = (unsigned)(a.H + R - 1) / R;
  for (unsigned u = blockIdx.x * 256u + threadIdx.x; u < units; u += gridDim.x * 256u) {
    const int c8 = (int)(u % (unsigned)a.C8);
    unsigned r = u / (unsigned)a.C8;
    const int x = (int)(r % (unsigned)a.W);
    r /= (unsigned)a.W;
    const int y0 = R * (int)(r % H4), n = (int)(r / H4);
    const unsigned short* dn = a.dy + (long long)n * a.dy_sn;
    const long long xo = (long long)n * a.x_sn + (long long)y0 * a.x_sh + (long long)x * a.x_sw + c8 * 8;
    const long long go = (long long)n * a.g_sn + (long long)y0 * a.g_sh + (long long)x * a.g_sw + c8 * 8;
    u32x4 xv[R], gv[R];                                     // the mask operand (and G), requested now and used after the taps
#pragma unroll
    for (int i = 0; i < R; ++i) {
      xv[i] = *reinterpret_cast<const u32x4*>(a.x + xo + (long long)(y0 + i < a.H ? i : 0) * a.x_sh);
      if constexpr (ACC1) gv[i] = *reinterpret_cast<const u32x4*>(a.g + go + (long long)(y0 + i < a.H ? i : 0) * a.g_sh);
    }
    f32x2 da[R][4];
#pragma unroll
    for (int i = 0; i < R; ++i)
#pragma unroll
      for (int k = 0; k < 4; ++k) da[i][k] = f32x2{0.f, 0.f};
    // window row wr = 0 .. R + KS - 2 is dy row y0 + pad - (KS - 1) + wr; pixel i, filter row ky reads window row i + KS - 1 - ky
#pragma unroll 1
    for (int kx = 0; kx < KS; ++kx) {
      const int ox = x + a.pad - kx;
      const bool cok = ox >= 0 && ox < a.Wo;
      float dv[R + KS - 1];
#pragma unroll
      for (int wr = 0; wr < R + KS - 1; ++wr) {
        const int oy = y0 + a.pad - (KS - 1) + wr;
        const bool ok = cok && oy >= 0 && oy < a.Ho;      // (clamped address, zeroed after: see conv_cout1_kernel's load; this one had 28
        const float v = fd_cvt1<FmtG>(dn[ok ? oy * a.dy_sh + ox * a.dy_sw : 0]);      // dependent 2-byte loads per unit, each waited for)
        dv[wr] = ok ? v : 0.f;
      }
#pragma unroll
      for (int ky = 0; ky < KS; ++ky) {
        const float* wt = wl + (((ky * KS + kx) * 2) * a.C8 + c8) * 4;
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(wt), w1 = *reinterpret_cast<const f32x4*>(wt + a.C8 * 4);
        const f32x2 wa = {w0[0], w0[1]}, wb = {w0[2], w0[3]}, wc = {w1[0], w1[1]}, wd = {w1[2], w1[3]};
#pragma unroll
        for (int i = 0; i < R; ++i) {
          const float e = dv[i + KS - 1 - ky];
          const f32x2 d2 = {e, e};
          da[i][0] = __builtin_elementwise_fma(d2, wa, da[i][0]), da[i][1] = __builtin_elementwise_fma(d2, wb, da[i][1]);
          da[i][2] = __builtin_elementwise_fma(d2, wc, da[i][2]), da[i][3] = __builtin_elementwise_fma(d2, wd, da[i][3]);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const f32x8 fx = fd_cvt8<FmtA>(xv[i]);
      unsigned short* gp = a.g + go + (long long)i * a.g_sh;
      f32x8 o = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if constexpr (ACC1) o = fd_cvt8<FmtG>(gv[i]);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] += (c8 * 8 + e < a.C) ? da[i][e >> 1][e & 1] * (fx[e] > 0.f ? 1.f : a.slope) : 0.f;
      if (y0 + i < a.H) *reinterpret_cast<u32x4*>(gp) = fd_pk8<FmtG>(o);
    }
  }
}

// ---- weight gradient of the one-filter conv: dW[c][ky][kx] = sum over output pixels of dy[p] * a[p + (ky, kx) - pad][c] ----------------
// On the matrix-pipe kernels the single output channel is a 32-wide tile with one live row (conv_wgrad4x4_tr: 150 us at B = 16 @
// 127 x 127 x 288 for 149 MB of input, 19 us of HBM time).  Read the other way round it is a streaming reduction: every input pixel
// a[iy][ix][c] meets the KS x KS gradient values around it -- dW[c][ky][kx] += a[iy][ix][c] * dy[iy + pad - ky][ix + pad - kx] -- so a
// thread owns 8 channels, walks input pixels and keeps all KS^2 x 8 sums in registers: one 16-byte load and KS^2 x 4 packed FMAs per
// pixel.  Workgroup = (image, band of W1_ROWS input rows), 256 threads = C / 8 channel groups x 256 / (C / 8) pixel slots; the band's
// dy rows sit in LDS as fp32 with a zero frame (no bounds tests in the tap loop); the slots are summed through LDS in a fixed order
// and the workgroup leaves ONE partial [C][KS][KS], which wgrad_reduce_kernel sums over the workgroups.
#ifndef W1_PF
#define W1_PF 2
#endif
#ifndef W1_ROWS_N
#define W1_ROWS_N 4
#endif
constexpr int W1_ROWS = W1_ROWS_N;
struct C1WgArgs {
  const unsigned short* x;     // NHWC fp16 view of the conv's input
  long long x_sn;
  int x_sh, x_sw, H, W, C, C8;
  const unsigned short* dy;    // NHWC bf16 view, channel 0 used
  long long dy_sn;
  int dy_sh, dy_sw, Ho, Wo, pad, bands;
  int pro_mode;
  float slope, eps;
  const float *mean, *var, *gamma, *beta;
  float* part;                 // [workgroup][C][KS][KS]
};

template <int KS>
__global__ __launch_bounds__(256) void wgrad_cout1_kernel(C1WgArgs a) {
  constexpr int KK = KS * KS, DR = W1_ROWS + KS - 1;
  extern __shared__ __attribute__((aligned(16))) char c1_lds[];
  const int dw_ = a.Wo + 2 * (KS - 1);                       // dy band row width with the zero frame
  float* dyl = reinterpret_cast<float*>(c1_lds);              // [DR][dw_]
  float* red = dyl;                                           // [slots][C8 * 8][KS] (after the pixel loop, over the band)
  const int tid = threadIdx.x;
  const int band = blockIdx.x % a.bands, n = blockIdx.x / a.bands;
  const int iy0 = band * W1_ROWS;
  const int nslot = 256 / a.C8;
  const int c8 = tid % a.C8, slot = tid / a.C8;
  const bool live = slot < nslot;
  // dy band: rows oy = iy0 + pad - (KS - 1) .. iy0 + pad + W1_ROWS - 1, columns ox = -(KS - 1) .. Wo + KS - 2, zeros outside
  const unsigned short* dn = a.dy + (long long)n * a.dy_sn;
  for (int i = tid; i < DR * dw_; i += 256) {
    const int r = i / dw_, cc = i - r * dw_;
    const int oy = iy0 + a.pad - (KS - 1) + r, ox = cc - (KS - 1);
    const bool ok = oy >= 0 && oy < a.Ho && ox >= 0 && ox < a.Wo;
    const float v = fd_cvt1<FmtG>(dn[ok ? oy * a.dy_sh + ox * a.dy_sw : 0]);      // (unconditional load, clamped address)
    dyl[i] = ok ? v : 0.f;
  }
  typedef f32x2 f2;
  f2 sc[4], sh[4];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = c8 * 8 + e, cc = min(c, a.C - 1);
    float s_ = 1.f, h_ = 0.f;
    if (a.pro_mode == 2) {
      const float g = a.gamma ? a.gamma[cc] : 1.f, b = a.beta ? a.beta[cc] : 0.f;
      s_ = g / sqrtf(a.var[cc] + a.eps);
      h_ = b - a.mean[cc] * s_;
    }
    sc[e >> 1][e & 1] = c < a.C ? s_ : 0.f, sh[e >> 1][e & 1] = c < a.C ? h_ : 0.f;
  }
  __syncthreads();
  f2 acc[KK][4];
#pragma unroll
  for (int t = 0; t < KK; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[t][q] = f2{0.f, 0.f};
  const int rows = min(W1_ROWS, a.H - iy0);
  const int npix = rows * a.W;
  const unsigned short* xn = a.x + (long long)n * a.x_sn + c8 * 8;
  if (live) {
    for (int p0 = slot; p0 < npix; p0 += W1_PF * nslot) {      // W1_PF pixels in flight per thread
      u32x4 xv[W1_PF];
      int ry[W1_PF], rx[W1_PF];
      bool ok[W1_PF];
#pragma unroll
      for (int k = 0; k < W1_PF; ++k) {
        const int p = p0 + k * nslot;
        ok[k] = p < npix;
        const int pc = ok[k] ? p : p0;
        ry[k] = pc / a.W, rx[k] = pc - ry[k] * a.W;
        xv[k] = *reinterpret_cast<const u32x4*>(xn + (long long)(iy0 + ry[k]) * a.x_sh + (long long)rx[k] * a.x_sw);
      }
#pragma unroll
      for (int k = 0; k < W1_PF; ++k) {
        const f32x8 f = fd_cvt8<FmtA>(xv[k]);
        const float lv = ok[k] ? 1.f : 0.f;
        f2 av[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f2 t = __builtin_elementwise_fma(f2{f[2 * q], f[2 * q + 1]}, sc[q], sh[q]);
          if (a.pro_mode != 0) t = __builtin_elementwise_max(t, t * a.slope);
          av[q] = t * lv;
        }
        // input pixel (ry, rx) of the band meets dy[iy + pad - ky][ix + pad - kx]: band row ry + (KS - 1) - ky, framed column rx + pad - kx + (KS - 1)
        const float* dp = dyl + (ry[k] + KS - 1) * dw_ + rx[k] + a.pad + (KS - 1);
#pragma unroll
        for (int ky = 0; ky < KS; ++ky)
#pragma unroll
          for (int kx = 0; kx < KS; ++kx) {
            const float d = dp[-ky * dw_ - kx];
            const f2 d2 = {d, d};
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[ky * KS + kx][q] = __builtin_elementwise_fma(d2, av[q], acc[ky * KS + kx][q]);
          }
      }
    }
  }
  // the slots' sums, KS taps (one filter row) at a time through LDS -- all KS^2 at once would be 129 KB for C = 288 and leave room for
  // one workgroup (4 waves) per CU; the dy band is dead by now
  const int cp = a.C8 * 8;
  float* out = a.part + (long long)blockIdx.x * a.C * KK;
#pragma unroll
  for (int ky = 0; ky < KS; ++ky) {
    __syncthreads();
    if (live)
#pragma unroll
      for (int kx = 0; kx < KS; ++kx)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          red[(slot * cp + c8 * 8 + 2 * q) * KS + kx] = acc[ky * KS + kx][q][0];
          red[(slot * cp + c8 * 8 + 2 * q + 1) * KS + kx] = acc[ky * KS + kx][q][1];
        }
    __syncthreads();
    for (int i = tid; i < a.C * KS; i += 256) {               // i = c * KS + kx
      float t = 0.f;
      for (int s_ = 0; s_ < nslot; ++s_) t += red[s_ * cp * KS + i];
      const int c = i / KS, kx = i - c * KS;
      out[c * KK + ky * KS + kx] = t;                          // dW[0][c][ky][kx] order
    }
  }
}

}  // namespace

// forward: Cout == 1 exactly, 4x4 (or 3x3) stride 1, NCHW fp32 output, no bias, no upsample
bool conv_cout1_fits(const ConvArgs& a, int cout_total, int ksize, int stride, bool pool) {
  return cout_total == 1 && a.Cout == 1 && (ksize == 4 || ksize == 3) && stride == 1 && !pool && a.out_nchw_f32 && a.bias == nullptr && !a.upsample &&
         !a.grad_io && a.stats == nullptr && a.Cin % 8 == 0 && a.y_sw == 1 && FD_TUNE_GETENV("FDGAN_DEBUG_NO_C1") == nullptr;
}

int conv_cout1_launch(const ConvArgs& a, long long nimg, int ksize, FdConvInfo* info, bool dry, hipStream_t stream) {
  C1FwdArgs c{};
  c.x = a.x, c.x_sn = a.x_sn, c.x_sh = a.x_sh, c.x_sw = a.x_sw, c.H = a.Hs, c.W = a.Ws, c.Cin = a.Cin, c.nchunk = a.nchunk;
  c.w = a.w, c.pro_mode = a.pro_mode, c.slope = a.p_slope, c.eps = a.eps;
  c.mean = a.p_mean, c.var = a.p_var, c.gamma = a.p_gamma, c.beta = a.p_beta;
  c.y = static_cast<float*>(a.y), c.y_sn = a.y_sn, c.y_sh = a.y_sh, c.Ho = a.Ho, c.Wo = a.Wo, c.pad = a.pad, c.e_slope = a.e_slope;
  // the matrix-pipe form (taps as output channels) for up to 512 input channels; FDGAN_DEBUG_C1_OLD (tuning builds) keeps the dot-product kernel
  if (a.Cin <= 512 && a.Wo >= 1 && FD_TUNE_GETENV("FDGAN_DEBUG_C1_OLD") == nullptr) {
    const int nks = a.Cin <= 320 ? 5 : 8, cstep = C1M_CW - (ksize - 1);
    c.tiles_x = (a.Wo - 1 - a.pad < 0 ? 0 : (a.Wo - 1 - a.pad) / cstep) + 1;      // column blocks: window starts -pad .. Wo - 1 - pad
    c.tiles_y = (a.Ho + C1M_R - 1) / C1M_R;
    const unsigned lds_m = (unsigned)((C1M_R + ksize - 1) * 16 * C1M_CW * 4 + nks * 2 * 1024 + nks * 64 * 8);
    dim3 grid_m((unsigned)(nimg * c.tiles_x * c.tiles_y));
    if (info) {
      info->stats_rows = 0, info->stats_cpad = 0, info->grid_x = grid_m.x, info->grid_y = 1, info->lds_bytes = lds_m;
    }
    if (dry) return FD_OK;
    static bool attr_m = false;      // (per process like every launcher's: the library drives one device per process)
    if (!attr_m) {
      const void* ks[4] = {reinterpret_cast<const void*>(&conv_cout1_mfma_kernel<4, 5>), reinterpret_cast<const void*>(&conv_cout1_mfma_kernel<4, 8>),
                           reinterpret_cast<const void*>(&conv_cout1_mfma_kernel<3, 5>), reinterpret_cast<const void*>(&conv_cout1_mfma_kernel<3, 8>)};
      for (const void* k : ks) {
        hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) FD_FAIL(FD_ELAUNCH, "hipFuncSetAttribute(conv_cout1_mfma): %s", hipGetErrorString(e));
      }
      attr_m = true;
    }
    if (ksize == 4) return nks == 5 ? fd_launch(&conv_cout1_mfma_kernel<4, 5>, "conv4x4_cout1", grid_m, dim3(512), lds_m, c, stream)
                                    : fd_launch(&conv_cout1_mfma_kernel<4, 8>, "conv4x4_cout1", grid_m, dim3(512), lds_m, c, stream);
    return nks == 5 ? fd_launch(&conv_cout1_mfma_kernel<3, 5>, "conv3x3_cout1", grid_m, dim3(512), lds_m, c, stream)
                    : fd_launch(&conv_cout1_mfma_kernel<3, 8>, "conv3x3_cout1", grid_m, dim3(512), lds_m, c, stream);
  }
  c.tiles_x = (a.Wo + C1_TW - 1) / C1_TW, c.tiles_y = (a.Ho + C1_TH - 1) / C1_TH;
  const int ih = C1_TH + ksize - 1, iw = C1_TW + ksize - 1, npixr = (ih * iw + 15) / 16 * 16;
  const unsigned lds = 2u * 4 * npixr * 16 + 2u * ksize * ksize * 64 + a.nchunk * 32 * 8;
  dim3 grid((unsigned)(nimg * c.tiles_x * c.tiles_y));
  if (info) {
    info->stats_rows = 0, info->stats_cpad = 0, info->grid_x = grid.x, info->grid_y = 1, info->lds_bytes = lds;
  }
  if (dry) return FD_OK;
  static bool attr_done = false;
  if (!attr_done) {
    for (const void* k : {reinterpret_cast<const void*>(&conv_cout1_kernel<4>), reinterpret_cast<const void*>(&conv_cout1_kernel<3>)}) {
      hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) FD_FAIL(FD_ELAUNCH, "hipFuncSetAttribute(conv_cout1): %s", hipGetErrorString(e));
    }
    attr_done = true;
  }
  if (ksize == 4) return fd_launch(&conv_cout1_kernel<4>, "conv4x4_cout1", grid, dim3(256), lds, c, stream);
  return fd_launch(&conv_cout1_kernel<3>, "conv3x3_cout1", grid, dim3(256), lds, c, stream);
}

// data gradient of a stride-1 conv with ONE forward filter behind an activation-only prologue: returns 1 when the shape is not this kernel's
int dgrad_cout1_launch(const FdTensor* dy, const void* w_packed_flipped, const FdTensor* fwd_x, const FdPrologue* fwd_pro, const FdTensor* dpre,
                       int accumulate, const FdConvDesc* d, hipStream_t stream) {
  const bool norm = fwd_pro && fwd_pro->mean;
  const int pad_fwd = d->ksize - 1 - d->pad;
  if (dy->c != 1 || norm || d->stride != 1 || d->ksize < 2 || d->ksize > 4 || pad_fwd < 0 || dpre->c % 8 != 0 ||
      dpre->h != dy->h + d->ksize - 1 - 2 * pad_fwd || dpre->w != dy->w + d->ksize - 1 - 2 * pad_fwd || FD_TUNE_GETENV("FDGAN_DEBUG_NO_C1") != nullptr)
    return 1;
  const int act = fwd_pro ? fwd_pro->act : FD_ACT_NONE;
  C1BwdArgs c{};
  c.dy = static_cast<const unsigned short*>(dy->ptr), c.dy_sn = dy->stride[0], c.dy_sh = (int)dy->stride[1], c.dy_sw = (int)dy->stride[2];
  c.Ho = (int)dy->h, c.Wo = (int)dy->w;
  c.w = static_cast<const unsigned short*>(w_packed_flipped);
  c.x = static_cast<const unsigned short*>(fwd_x->ptr), c.x_sn = fwd_x->stride[0], c.x_sh = (int)fwd_x->stride[1], c.x_sw = (int)fwd_x->stride[2];
  c.g = static_cast<unsigned short*>(dpre->ptr), c.g_sn = dpre->stride[0], c.g_sh = (int)dpre->stride[1], c.g_sw = (int)dpre->stride[2];
  c.H = (int)dpre->h, c.W = (int)dpre->w, c.C = (int)dpre->c, c.C8 = (int)((dpre->c + 7) / 8), c.ks = d->ksize, c.pad = pad_fwd, c.acc = accumulate;
  c.slope = act == FD_ACT_RELU ? 0.f : (act == FD_ACT_LEAKY02 ? 0.2f : 1.f);
  c.total = dpre->n * ((dpre->h + 3) / 4) * dpre->w * c.C8;                    // units: (image, four rows, pixel, 8-channel piece)
  c.ntile = (int)((dpre->c + 15) / 16);
  const unsigned lds = (unsigned)(d->ksize * d->ksize * c.C8 * 8 * 4);
  if (lds > 64 * 1024 || c.total >= (1ll << 31) || dy->n * dy->stride[0] >= (1ll << 31)) return 1;
  const long long nb = (c.total + 255) / 256;
  const dim3 grid((unsigned)(nb < 2048 ? nb : 2048));
  const bool acc1 = accumulate == 1;
  switch (d->ksize) {
    case 4: return acc1 ? fd_launch(&dgrad_cout1_kernel<4, true>, "dgrad_cout1", grid, dim3(256), lds, c, stream)
                        : fd_launch(&dgrad_cout1_kernel<4, false>, "dgrad_cout1", grid, dim3(256), lds, c, stream);
    case 3: return acc1 ? fd_launch(&dgrad_cout1_kernel<3, true>, "dgrad_cout1", grid, dim3(256), lds, c, stream)
                        : fd_launch(&dgrad_cout1_kernel<3, false>, "dgrad_cout1", grid, dim3(256), lds, c, stream);
    default: return acc1 ? fd_launch(&dgrad_cout1_kernel<2, true>, "dgrad_cout1", grid, dim3(256), lds, c, stream)
                         : fd_launch(&dgrad_cout1_kernel<2, false>, "dgrad_cout1", grid, dim3(256), lds, c, stream);
  }
}

// weight gradient of a stride-1 conv with ONE filter: 0 launched (partials in `workspace`: *nsplit_out arrays of cin * k * k floats),
// 1 the shape is not this kernel's (nothing launched), < 0 error
int wgrad_cout1_launch(const FdTensor* x, const FdTensor* dy, int ksize, int stride, int pad, int pro_mode, float slope, float eps,
                       const float* mean, const float* var, const float* gamma, const float* beta, float* workspace,
                       long long workspace_floats, long long* nsplit_out, hipStream_t stream) {
  if (stride != 1 || ksize < 2 || ksize > 4 || x->c % 8 != 0 || x->c < 8 || x->c > 2048 || workspace == nullptr ||
      FD_TUNE_GETENV("FDGAN_DEBUG_NO_C1") != nullptr)
    return 1;
  C1WgArgs c{};
  c.x = static_cast<const unsigned short*>(x->ptr), c.x_sn = x->stride[0], c.x_sh = (int)x->stride[1], c.x_sw = (int)x->stride[2];
  c.H = (int)x->h, c.W = (int)x->w, c.C = (int)x->c, c.C8 = (int)(x->c / 8);
  c.dy = static_cast<const unsigned short*>(dy->ptr), c.dy_sn = dy->stride[0], c.dy_sh = (int)dy->stride[1], c.dy_sw = (int)dy->stride[2];
  c.Ho = (int)dy->h, c.Wo = (int)dy->w, c.pad = pad;
  c.bands = (c.H + W1_ROWS - 1) / W1_ROWS;
  c.pro_mode = pro_mode, c.slope = slope, c.eps = eps, c.mean = mean, c.var = var, c.gamma = gamma, c.beta = beta;
  if (c.C8 > 256) return 1;
  const int nslot = 256 / c.C8, kk = ksize * ksize, dr = W1_ROWS + ksize - 1, dwid = c.Wo + 2 * (ksize - 1);
  const long long lds_band = (long long)dr * dwid * 4, lds_red = (long long)nslot * c.C8 * 8 * ksize * 4;
  const long long lds = lds_band > lds_red ? lds_band : lds_red;
  const long long nwg = (long long)x->n * c.bands;
  if (lds > 160 * 1024 || nwg >= (1ll << 31) || nwg * c.C * kk > workspace_floats || dy->n * dy->stride[0] >= (1ll << 31)) return 1;
  c.part = workspace;
  *nsplit_out = nwg;
  static bool attr_done = false;
  if (!attr_done) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_cout1_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_cout1_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_cout1_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  const dim3 grid((unsigned)nwg);
  switch (ksize) {
    case 4: return fd_launch(&wgrad_cout1_kernel<4>, "wgrad_cout1", grid, dim3(256), (unsigned)lds, c, stream);
    case 3: return fd_launch(&wgrad_cout1_kernel<3>, "wgrad_cout1", grid, dim3(256), (unsigned)lds, c, stream);
    default: return fd_launch(&wgrad_cout1_kernel<2>, "wgrad_cout1", grid, dim3(256), (unsigned)lds, c, stream);
  }
}
