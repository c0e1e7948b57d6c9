// conv_igemm.hip -- NHWC bf16 implicit-GEMM convolution on gfx950 MFMA.
//
//   y = act_e( conv( pool?( act_p( bn?(x) ) ) ) + bias )  [nearest x2]  (+ batch statistics)
//
// GEMM view: D[cout][pixel] = sum_k W[cout][k] * A[k][pixel], k = (tap, cin).
// The filter is the MFMA "A" operand and the pixels the "B" operand, so that a lane
// of the v_mfma_f32_16x16x32_bf16 result holds 4 CONSECUTIVE output channels of one
// pixel (an 8-byte NHWC store) instead of 4 pixels of one channel.
//
// One workgroup = TH x 16 output pixels of one image x BN output channels:
//   * the input halo tile ((TH-1)*S+KS) x (15*S+KS) pixels x 32 channels is staged
//     ONCE per 32-channel chunk into LDS -- global -> registers -> (BN affine, ReLU,
//     2x2 average, zero padding) -> ds_write_b128 -- and re-used by all KS*KS taps;
//   * LDS image: 4 planes (one per 8-channel group) of [pixel][8 ch] 16-byte slots,
//     plane stride a multiple of 256 B, so the ds_read_b128 fragment reads of a
//     16-pixel row are conflict-free for every tap shift (MI355X_MICROARCH LDS table);
//   * the filter arrives pre-packed in fragment order (1 KiB per 16 cout x 32 k), so
//     its LDS image is lane-linear: lane l reads its fragment at base + 16*l;
//   * both stagings are double-buffered: loads for step s+1 are issued before the
//     MFMAs of step s and written to LDS after them (one barrier per step).
//
// Reference call sites replaced: see include/fdgan_hip.h (fdgan_conv2d_fwd).
#pragma once
#include "common.h"

struct ConvArgs {
  const unsigned short* x;
  long long x_sn;
  int x_sh, x_sw;  // element strides of h, w of the SOURCE (pre-pool) tensor
  int Hs, Ws;      // source spatial size
  int Cin;         // logical input channels
  int Cin8;        // readable 8-channel groups = ceil(Cin/8)
  int nchunk;      // ceil(Cin/32)
  const unsigned short* w;
  int ntile_total;  // ceil(Cout/16)
  const float* bias;
  // prologue
  int pro_mode;  // 0 raw, 1 activation only, 2 affine + activation
  int p_act;
  const float *p_mean, *p_var, *p_gamma, *p_beta;
  float eps, momentum, unbias;
  float *run_mean, *run_var;
  long long* nbt;
  // output
  void* y;
  long long y_sn, y_sc;
  int y_sh, y_sw;
  int Ho, Wo;  // conv output size (before the optional x2 upsample)
  int Cout;    // channels to store (view->c)
  int CoutW;   // the filter's true output channels (bias length)
  int e_act, out_nchw_f32, upsample;
  float* stats;
  int stats_cpad;
  int tiles_x, tiles_y;
  int pad;
};

__device__ __forceinline__ u32x4 lds_read16(const char* p) { return *reinterpret_cast<const u32x4*>(p); }
__device__ __forceinline__ void lds_write16(char* p, u32x4 v) { *reinterpret_cast<u32x4*>(p) = v; }

// bf16x8 (as 4 dwords) -> act(x*sc+sh) -> bf16x8.  sc/sh point at 8 floats in LDS.
__device__ __forceinline__ f32x8 fd_affine_act(u32x4 raw, const float* sc, const float* sh, int mode, int act) {
  f32x8 f = __builtin_convertvector(__builtin_bit_cast(bf16x8, raw), f32x8);
  if (mode == 2) {
    f32x4 s0 = *reinterpret_cast<const f32x4*>(sc), s1 = *reinterpret_cast<const f32x4*>(sc + 4);
    f32x4 h0 = *reinterpret_cast<const f32x4*>(sh), h1 = *reinterpret_cast<const f32x4*>(sh + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      f[e] = fmaf(f[e], s0[e], h0[e]);
      f[e + 4] = fmaf(f[e + 4], s1[e], h1[e]);
    }
  }
  if (act == FD_ACT_RELU) {
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = fmaxf(f[e], 0.f);
  } else if (act == FD_ACT_LEAKY02) {
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = fmaxf(f[e], 0.2f * f[e]);
  }
  return f;
}
__device__ __forceinline__ u32x4 fd_pack8(f32x8 f) {
  return __builtin_bit_cast(u32x4, __builtin_convertvector(f, bf16x8));
}

template <int KS, int STRIDE, int POOL, int PT, int CT, int WM, int WN, int TPS>
struct ConvCfg {
  static constexpr int NT = 64 * WM * WN;
  static constexpr int TH = WM * PT, TW = 16;
  static constexpr int IH = (TH - 1) * STRIDE + KS, IW = (TW - 1) * STRIDE + KS;
  static constexpr int NPIX = IH * IW;
  static constexpr int NPIXR = (NPIX + 15) / 16 * 16;
  static constexpr int PLANE_B = NPIXR * 16;
  static constexpr int IN_BYTES = 4 * PLANE_B;
  static constexpr int CTB = WN * CT;
  static constexpr int BN = CTB * 16;
  static constexpr int KK = KS * KS;
  static constexpr int NSPC = KK / TPS;  // steps per chunk
  static constexpr int W_BYTES = TPS * CTB * 1024;
  static constexpr int IN_UNITS = 4 * NPIXR;
  static constexpr int IN_UPT = (IN_UNITS + NT - 1) / NT;
  static constexpr int W_UNITS = TPS * CTB * 64;
  static constexpr int W_UPT = (W_UNITS + NT - 1) / NT;
  static constexpr int NLOAD = POOL ? 4 : 1;
  static_assert(KK % TPS == 0, "taps per stage must divide KS*KS");
  static_assert(!POOL || (KS == 1 && STRIDE == 1), "pool prologue is for 1x1 convs");
  static unsigned lds_bytes(int nchunk) { return 2 * IN_BYTES + 2 * W_BYTES + nchunk * 32 * 8; }
};

template <int KS, int STRIDE, int POOL, int PT, int CT, int WM, int WN, int TPS>
__global__ __launch_bounds__(64 * WM * WN, 2) void conv_igemm_kernel(ConvArgs a) {
  using C = ConvCfg<KS, STRIDE, POOL, PT, CT, WM, WN, TPS>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* in_lds = smem;                      // [2][IN_BYTES]
  char* w_lds = smem + 2 * C::IN_BYTES;     // [2][W_BYTES]
  float* sc_lds = reinterpret_cast<float*>(smem + 2 * C::IN_BYTES + 2 * C::W_BYTES);  // [nchunk*32]
  float* sh_lds = sc_lds + a.nchunk * 32;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int m = lane & 15, kgl = lane >> 4;

  int tile = blockIdx.x;
  const int tx = tile % a.tiles_x;
  tile /= a.tiles_x;
  const int ty = tile % a.tiles_y;
  const int n = tile / a.tiles_y;
  const int by = blockIdx.y;
  const int oy0 = ty * C::TH, ox0 = tx * C::TW;

  // ---- preamble: fold BatchNorm into per-channel scale/shift (+ train-mode side effects)
  if (a.pro_mode == 2) {
    for (int c = tid; c < a.nchunk * 32; c += C::NT) {
      float sc = 0.f, sh = 0.f;
      if (c < a.Cin) {
        const float mean = a.p_mean[c], var = a.p_var[c];
        const float g = a.p_gamma ? a.p_gamma[c] : 1.f, b = a.p_beta ? a.p_beta[c] : 0.f;
        sc = g / sqrtf(var + a.eps);
        sh = b - mean * sc;
        if (a.run_mean != nullptr && blockIdx.x == 0 && blockIdx.y == 0) {
          a.run_mean[c] = (1.f - a.momentum) * a.run_mean[c] + a.momentum * mean;
          a.run_var[c] = (1.f - a.momentum) * a.run_var[c] + a.momentum * var * a.unbias;
        }
      }
      sc_lds[c] = sc;
      sh_lds[c] = sh;
    }
    if (a.nbt != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) *a.nbt += 1;
  }

  // ---- per-thread staging maps (fixed across the K loop)
  const unsigned short* xn = a.x + (long long)n * a.x_sn;
  int goff[C::IN_UPT];  // element offset of the unit's first source pixel, -1: outside the image
  int ukg[C::IN_UPT];   // 8-channel group of the unit, -1: unit does not exist
#pragma unroll
  for (int i = 0; i < C::IN_UPT; ++i) {
    const int u = tid + i * C::NT;
    const int kg = u / C::NPIXR, p = u - kg * C::NPIXR;
    const bool exists = (u < C::IN_UNITS) && (p < C::NPIX);
    const int py = p / C::IW, px = p - py * C::IW;
    int gy, gx;
    bool inb;
    if (POOL) {
      gy = 2 * (oy0 + py);
      gx = 2 * (ox0 + px);
      inb = (oy0 + py < a.Ho) && (ox0 + px < a.Wo);
    } else {
      gy = oy0 * STRIDE - a.pad + py;
      gx = ox0 * STRIDE - a.pad + px;
      inb = gy >= 0 && gy < a.Hs && gx >= 0 && gx < a.Ws;
    }
    ukg[i] = exists ? kg : -1;
    goff[i] = (exists && inb) ? gy * a.x_sh + gx * a.x_sw + kg * 8 : -1;
  }
  const unsigned short* wsrc[C::W_UPT];
#pragma unroll
  for (int i = 0; i < C::W_UPT; ++i) {
    const int j = tid + i * C::NT;
    const int t = j / (C::CTB * 64), rem = j - t * (C::CTB * 64);
    const int tl = rem >> 6, ln = rem & 63;
    const int tile16 = by * C::CTB + tl;
    wsrc[i] = (j < C::W_UNITS && tile16 < a.ntile_total)
                  ? a.w + ((long long)t * a.ntile_total + tile16) * 512 + ln * 8
                  : nullptr;
  }

  u32x4 rin[C::IN_UPT][C::NLOAD];
  u32x4 rw[C::W_UPT];
  const u32x4 zero4 = {0u, 0u, 0u, 0u};

  auto load_in = [&](int chunk) {
#pragma unroll
    for (int i = 0; i < C::IN_UPT; ++i) {
      const bool ok = goff[i] >= 0 && (chunk * 4 + ukg[i]) < a.Cin8;
      const unsigned short* p = xn + (ok ? goff[i] + chunk * 32 : 0);
      if (POOL) {
        rin[i][0] = ok ? *reinterpret_cast<const u32x4*>(p) : zero4;
        rin[i][1] = ok ? *reinterpret_cast<const u32x4*>(p + a.x_sw) : zero4;
        rin[i][2] = ok ? *reinterpret_cast<const u32x4*>(p + a.x_sh) : zero4;
        rin[i][3] = ok ? *reinterpret_cast<const u32x4*>(p + a.x_sh + a.x_sw) : zero4;
      } else {
        rin[i][0] = ok ? *reinterpret_cast<const u32x4*>(p) : zero4;
      }
    }
  };
  auto store_in = [&](char* buf, int chunk) {
#pragma unroll
    for (int i = 0; i < C::IN_UPT; ++i) {
      if (ukg[i] < 0) continue;
      const bool ok = goff[i] >= 0 && (chunk * 4 + ukg[i]) < a.Cin8;
      u32x4 v = zero4;
      if (ok) {
        const float* sc = sc_lds + chunk * 32 + ukg[i] * 8;
        const float* sh = sh_lds + chunk * 32 + ukg[i] * 8;
        if (POOL) {
          f32x8 f = fd_affine_act(rin[i][0], sc, sh, a.pro_mode, a.p_act);
          f += fd_affine_act(rin[i][1], sc, sh, a.pro_mode, a.p_act);
          f += fd_affine_act(rin[i][2], sc, sh, a.pro_mode, a.p_act);
          f += fd_affine_act(rin[i][3], sc, sh, a.pro_mode, a.p_act);
          v = fd_pack8(f * 0.25f);
        } else if (a.pro_mode != 0) {
          v = fd_pack8(fd_affine_act(rin[i][0], sc, sh, a.pro_mode, a.p_act));
        } else {
          v = rin[i][0];
        }
      }
      lds_write16(buf + (tid + i * C::NT) * 16, v);
    }
  };
  auto load_w = [&](int chunk, int tg) {
    const long long off = ((long long)chunk * C::KK + tg * TPS) * a.ntile_total * 512;
#pragma unroll
    for (int i = 0; i < C::W_UPT; ++i)
      rw[i] = wsrc[i] ? *reinterpret_cast<const u32x4*>(wsrc[i] + off) : zero4;
  };
  auto store_w = [&](char* buf) {
#pragma unroll
    for (int i = 0; i < C::W_UPT; ++i) {
      const int j = tid + i * C::NT;
      if (j < C::W_UNITS) lds_write16(buf + j * 16, rw[i]);
    }
  };

  f32x4 acc[PT][CT];
#pragma unroll
  for (int p = 0; p < PT; ++p)
#pragma unroll
    for (int c = 0; c < CT; ++c) acc[p][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nsteps = a.nchunk * C::NSPC;
  __syncthreads();  // scale/shift visible
  load_in(0);
  load_w(0, 0);
  store_in(in_lds, 0);
  store_w(w_lds);
  __syncthreads();

  // fragment base addresses
  const char* xfrag0 = in_lds + kgl * C::PLANE_B + ((wm * PT * STRIDE) * C::IW + m * STRIDE) * 16;
  const char* wfrag0 = w_lds + ((wn * CT) * 64 + lane) * 16;

  for (int s = 0; s < nsteps; ++s) {
    const int chunk = s / C::NSPC, tg = s - chunk * C::NSPC;
    const bool has_next = (s + 1) < nsteps;
    const int chunk1 = (s + 1) / C::NSPC, tg1 = (s + 1) - chunk1 * C::NSPC;
    const bool new_chunk = has_next && (tg1 == 0);
    if (has_next) load_w(chunk1, tg1);
    if (new_chunk) load_in(chunk1);

    const char* xb = xfrag0 + (chunk & 1) * C::IN_BYTES;
    const char* wb = wfrag0 + (s & 1) * C::W_BYTES;
#pragma unroll
    for (int t = 0; t < TPS; ++t) {
      int dy, dx;
      if (TPS == C::KK) {  // compile-time tap
        dy = t / KS;
        dx = t % KS;
      } else {
        const int tap = tg * TPS + t;
        dy = tap / KS;
        dx = tap - dy * KS;
      }
      bf16x8 wf[CT], xf[PT];
#pragma unroll
      for (int c = 0; c < CT; ++c)
        wf[c] = __builtin_bit_cast(bf16x8, lds_read16(wb + (t * C::CTB + c) * 1024));
#pragma unroll
      for (int p = 0; p < PT; ++p)
        xf[p] = __builtin_bit_cast(bf16x8, lds_read16(xb + ((p * STRIDE + dy) * C::IW + dx) * 16));
#pragma unroll
      for (int p = 0; p < PT; ++p)
#pragma unroll
        for (int c = 0; c < CT; ++c)
          acc[p][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[c], xf[p], acc[p][c], 0, 0, 0);
    }

    if (has_next) store_w(w_lds + ((s + 1) & 1) * C::W_BYTES);
    if (new_chunk) store_in(in_lds + (chunk1 & 1) * C::IN_BYTES, chunk1);
    __syncthreads();
  }

  // ---- epilogue: bias, activation, batch statistics, store
  const int col = ox0 + m;
  float ssum[CT][4], ssq[CT][4];
#pragma unroll
  for (int c = 0; c < CT; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) ssum[c][r] = ssq[c][r] = 0.f;

#pragma unroll
  for (int c = 0; c < CT; ++c) {
    const int cout0 = by * C::BN + (wn * CT + c) * 16 + kgl * 4;
    float bv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = (a.bias != nullptr && cout0 + r < a.CoutW) ? a.bias[cout0 + r] : 0.f;
#pragma unroll
    for (int p = 0; p < PT; ++p) {
      const int row = oy0 + wm * PT + p;
      const bool valid = row < a.Ho && col < a.Wo;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = fd_act(acc[p][c][r] + bv[r], a.e_act);
        if (valid) {
          ssum[c][r] += v[r];
          ssq[c][r] += v[r] * v[r];
        }
      }
      if (!valid || cout0 >= a.Cout) continue;
      if (a.out_nchw_f32) {
        float* yp = reinterpret_cast<float*>(a.y) + (long long)n * a.y_sn + (long long)cout0 * a.y_sc;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (cout0 + r < a.Cout) {
            if (a.upsample) {
              float* q = yp + r * a.y_sc + (long long)(2 * row) * a.y_sh + (2 * col) * a.y_sw;
              q[0] = v[r];
              q[a.y_sw] = v[r];
              q[a.y_sh] = v[r];
              q[a.y_sh + a.y_sw] = v[r];
            } else {
              yp[r * a.y_sc + (long long)row * a.y_sh + col * a.y_sw] = v[r];
            }
          }
      } else {
        unsigned short* yp = reinterpret_cast<unsigned short*>(a.y) + (long long)n * a.y_sn + cout0;
        typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
        typedef __attribute__((ext_vector_type(4))) float f4;
        const bf16x4 pk = __builtin_convertvector((f4){v[0], v[1], v[2], v[3]}, bf16x4);
        const u32x2 bits = __builtin_bit_cast(u32x2, pk);
        if (cout0 + 4 <= a.Cout) {
          if (a.upsample) {
            unsigned short* q = yp + (long long)(2 * row) * a.y_sh + (2 * col) * a.y_sw;
            *reinterpret_cast<u32x2*>(q) = bits;
            *reinterpret_cast<u32x2*>(q + a.y_sw) = bits;
            *reinterpret_cast<u32x2*>(q + a.y_sh) = bits;
            *reinterpret_cast<u32x2*>(q + a.y_sh + a.y_sw) = bits;
          } else {
            *reinterpret_cast<u32x2*>(yp + (long long)row * a.y_sh + col * a.y_sw) = bits;
          }
        } else {
          const unsigned short* hs = reinterpret_cast<const unsigned short*>(&bits);
          for (int r = 0; r < 4; ++r)
            if (cout0 + r < a.Cout) {
              if (a.upsample) {
                unsigned short* q = yp + r + (long long)(2 * row) * a.y_sh + (2 * col) * a.y_sw;
                q[0] = hs[r];
                q[a.y_sw] = hs[r];
                q[a.y_sh] = hs[r];
                q[a.y_sh + a.y_sw] = hs[r];
              } else {
                yp[r + (long long)row * a.y_sh + col * a.y_sw] = hs[r];
              }
            }
        }
      }
    }
  }

  if (a.stats != nullptr) {
    // reduce over the 16 pixels of a fragment row (lanes sharing lane>>4), then over waves
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s1 = ssum[c][r], s2 = ssq[c][r];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
          s1 += __shfl_xor(s1, o, 64);
          s2 += __shfl_xor(s2, o, 64);
        }
        ssum[c][r] = s1;
        ssq[c][r] = s2;
      }
    float* red = reinterpret_cast<float*>(smem);  // [WM*WN waves][CT*16][2]; K loop ended with a barrier
    if (m == 0) {
#pragma unroll
      for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int idx = (wave * CT * 16 + c * 16 + kgl * 4 + r) * 2;
          red[idx] = ssum[c][r];
          red[idx + 1] = ssq[c][r];
        }
    }
    __syncthreads();
    for (int cl = tid; cl < C::BN; cl += C::NT) {
      const int wn_ = cl / (CT * 16), idx = cl - wn_ * (CT * 16);
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int w_ = 0; w_ < WM; ++w_) {
        s1 += red[((w_ * WN + wn_) * CT * 16 + idx) * 2];
        s2 += red[((w_ * WN + wn_) * CT * 16 + idx) * 2 + 1];
      }
      float* dst = a.stats + ((long long)blockIdx.x * a.stats_cpad + by * C::BN + cl) * 2;
      dst[0] = s1;
      dst[1] = s2;
    }
  }
}

// ---------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------
// The dispatch table is written out explicitly: (ksize, stride, pool, width class).
//   narrow: BN = 32  (PT=4, CT=2, 4x1 waves, all taps of a chunk staged at once)
//   wide  : BN = 128 (PT=4, CT=8, 4x1 waves, one tap per stage for 3x3 / 4x4)
//   pool  : BN = 128, TH = 8 (PT=2): 4 source pixels per staged unit
#define FD_CONV_DISPATCH(KS_, ST_, POOL_, PT_, CT_, WM_, WN_, TPS_, NAME_)                                       \
  do {                                                                                                          \
    using C = ConvCfg<KS_, ST_, POOL_, PT_, CT_, WM_, WN_, TPS_>;                                                \
    a.tiles_x = (a.Wo + C::TW - 1) / C::TW;                                                                     \
    a.tiles_y = (a.Ho + C::TH - 1) / C::TH;                                                                     \
    dim3 grid((unsigned)(nimg * a.tiles_x * a.tiles_y), (unsigned)((cout_total + C::BN - 1) / C::BN), 1);      \
    dim3 block(C::NT, 1, 1);                                                                                    \
    a.stats_cpad = grid.y * C::BN;                                                                              \
    const unsigned lds = C::lds_bytes(a.nchunk);                                                                \
    if (info) {                                                                                                 \
      info->stats_rows = grid.x;                                                                                \
      info->stats_cpad = a.stats_cpad;                                                                          \
      info->grid_x = grid.x;                                                                                    \
      info->grid_y = grid.y;                                                                                    \
      info->lds_bytes = lds;                                                                                    \
    }                                                                                                           \
    if (dry) return FD_OK;                                                                                      \
    auto kfn = &conv_igemm_kernel<KS_, ST_, POOL_, PT_, CT_, WM_, WN_, TPS_>;                                   \
    static bool attr_done = false;                                                                              \
    if (!attr_done) {                                                                                           \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn),                                    \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);               \
      if (e != hipSuccess) FD_FAIL(FD_ELAUNCH, "hipFuncSetAttribute(%s): %s", NAME_, hipGetErrorString(e));     \
      attr_done = true;                                                                                         \
    }                                                                                                           \
    if (stats_cap >= 0 && (long long)grid.x * a.stats_cpad * 2 > stats_cap)                                     \
      FD_FAIL(FD_EINVAL, "stats workspace too small: need %lld floats, have %lld",                              \
              (long long)grid.x * a.stats_cpad * 2, stats_cap);                                                 \
    return fd_launch(kfn, NAME_, grid, block, lds, a, stream);                                                  \
  } while (0)


// one translation unit per kernel size (parallel compilation)
int conv_dispatch_k1(ConvArgs& a, long long nimg, int cout_total, int stride, bool pool, FdConvInfo* info,
                     long long stats_cap, bool dry, hipStream_t stream);
int conv_dispatch_k3(ConvArgs& a, long long nimg, int cout_total, int stride, bool pool, FdConvInfo* info,
                     long long stats_cap, bool dry, hipStream_t stream);
int conv_dispatch_k4(ConvArgs& a, long long nimg, int cout_total, int stride, bool pool, FdConvInfo* info,
                     long long stats_cap, bool dry, hipStream_t stream);
