// conv_igemm.h -- NHWC fp16 (forward) / bf16 (gradients) implicit-GEMM convolution on gfx950 MFMA.
//
//   y = act_e( conv( pool?( act_p( bn?(x) ) ) ) + bias )  [nearest x2]  (+ batch statistics)
//
// GEMM view: D[cout][pixel] = sum_k W[cout][k] * A[k][pixel], k = (tap, cin).
// The filter is the MFMA "A" operand and the pixels the "B" operand, so that a lane
// of the v_mfma_f32_16x16x32_{f16,bf16} result holds 4 CONSECUTIVE output channels of one
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
// Prologue and epilogue are branch-free: BatchNorm is an always-applied per-channel
// (scale, shift) pair (1, 0 without a norm) and ReLU / LeakyReLU / identity are
// max(v, slope*v) with slope 0 / 0.2 / 1.  tanh / sigmoid run as a separate
// elementwise pass (elementwise.hip) -- they only follow the two tiny final convs.
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
  float p_slope; // max(v, p_slope*v): 1 identity, 0 ReLU, 0.2 LeakyReLU
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
  float e_slope;
  int out_nchw_f32, upsample;
  float* stats;
  int stats_cpad;
  // backward-data epilogue (generic kernel only): the stored value is acc * act'(bn(fx)) with fx the FORWARD conv's raw
  // input at the output position, and the statistics are (sum v, sum v * fx) instead of (sum y, sum y^2)
  int grad_io;                  // x, the filter image and a 16-bit y are bf16 gradients (the MK instantiations); else fp16 activations
  int mk_mode;                  // 0 off (plain store), 1 activation only, 2 BatchNorm + activation
  const unsigned short* mk_x;   // NHWC fp16 (a forward activation), same n/h/w as y, >= Cout channels
  long long mk_sn;
  int mk_sh, mk_sw;
  float mk_slope, mk_eps;
  const float *mk_mean, *mk_var, *mk_gamma, *mk_beta;
  int mk_acc;                   // y is the gradient buffer of x: 1: y += gamma * rstd * dpre, 2: y = gamma * rstd * dpre (sole consumer); 0: y = dpre
  // in-kernel finalize by the last workgroup (kernels that set FdConvInfo.fused_finalize)
  float *fin_mean, *fin_var;
  unsigned* fin_counter;
  double fin_inv_count;
  int tiles_x, tiles_y;
  int seg_rows;  // conv3x3_rs: output rows per work item (tiles_x strips x tiles_y row segments per image)
  int pad;
  // x-stream 1x1 kernel (conv1x1_xs.hip)
  unsigned P;   // output pixels N*Ho*Wo
  int ntiles;   // pixel tiles walked by the persistent workgroups
  int nks;      // 64-channel k-steps
  int kgroup;   // k-steps of the filter resident in LDS at a time (>= nks: whole filter)
  int x_dense;  // x offset of pixel p is p * x_sw (no pooling, contiguous n/h/w)
  int y_dense;  // y offset of pixel p is p * y_sw (no upsample, contiguous n/h/w, NHWC)
  int y_vec16;  // NHWC 16-bit output, 16-byte aligned rows, no upsample: row stores allowed
  int dbg_skip;             // measurement aid: phases to skip (FDGAN_DEBUG_PHASES), 0 in production
  unsigned long long* dbg;  // measurement aid: per-wave phase cycle totals of workgroup 0 (or NULL)
};

__device__ __forceinline__ u32x4 lds_read16(const char* p) { return *reinterpret_cast<const u32x4*>(p); }
__device__ __forceinline__ void lds_write16(char* p, u32x4 v) { *reinterpret_cast<u32x4*>(p) = v; }

// 8 x 16-bit (as 4 dwords) -> max(t, slope*t), t = x*sc+sh.  sc/sh point at 8 floats in LDS.  F: element format of x (and of
// what the transform packs): FmtA in every forward kernel.
template <class F = FmtA>
__device__ __forceinline__ f32x8 fd_affine_act(u32x4 raw, const float* sc, const float* sh, float slope) {
  f32x8 f = fd_cvt8<F>(raw);
  const f32x4 s0 = *reinterpret_cast<const f32x4*>(sc), s1 = *reinterpret_cast<const f32x4*>(sc + 4);
  const f32x4 h0 = *reinterpret_cast<const f32x4*>(sh), h1 = *reinterpret_cast<const f32x4*>(sh + 4);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    f[e] = fmaf(f[e], s0[e], h0[e]);
    f[e + 4] = fmaf(f[e + 4], s1[e], h1[e]);
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) f[e] = fmaxf(f[e], slope * f[e]);
  return f;
}
template <class F = FmtA>
__device__ __forceinline__ u32x4 fd_pack8(f32x8 f) {
  return fd_pk8<F>(f);
}

// BatchNorm + ReLU on 8 channels in 20 VALU ops: 8 widening conversions, 4 v_pk_fma_f32,
// 4 v_cvt_pk_{f16,bf16}_f32 and the ReLU as 4 v_pk_max_i16 on the packed result (a negative fp16 / bf16 is a
// negative int16; rounding is monotone, so relu(round(t)) == round(relu(t))).
template <class F = FmtA, class FO = F>
__device__ __forceinline__ u32x4 fd_bn_relu8(u32x4 raw, const float* sc, const float* sh) {
  typedef f32x2 f32x2_t;
  typedef __attribute__((ext_vector_type(2))) short s16x2_t;
  const f32x4 s0 = *reinterpret_cast<const f32x4*>(sc), s1 = *reinterpret_cast<const f32x4*>(sc + 4);
  const f32x4 h0 = *reinterpret_cast<const f32x4*>(sh), h1 = *reinterpret_cast<const f32x4*>(sh + 4);
  const f32x2_t sv[4] = {{s0[0], s0[1]}, {s0[2], s0[3]}, {s1[0], s1[1]}, {s1[2], s1[3]}};
  const f32x2_t hv[4] = {{h0[0], h0[1]}, {h0[2], h0[3]}, {h1[0], h1[1]}, {h1[2], h1[3]}};
  u32x4 out;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f32x2_t f = fd_cvt2<F>(raw[i]);
    f = __builtin_elementwise_fma(f, sv[i], hv[i]);
    const s16x2_t pk = __builtin_bit_cast(s16x2_t, fd_pk2<FO>(f));
    out[i] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(pk, (s16x2_t){0, 0}));
  }
  return out;
}

// Register-operand forms (scale/shift already fetched from LDS): used where one lane applies the
// same 8 channels to several units, so the 4 LDS reads are paid once instead of per unit.
// (Round 6, measured and not adopted: the fp16 -> fp16 BatchNorm + ReLU as v_fma_mixlo_f16 / v_fma_mixhi_f16 -- fp16 source 0, fp32
// scale / shift, the rounded fp16 result written to one half of the destination: 12 instead of 20 VALU instructions per 8-channel
// fragment, conv1x1_ds 1-3 % faster, conv3x3_pw's spill gone.  Its rounding is not v_cvt_pk_f16_f32's: netG moved from 64.66 to
// 64.85 dB (train) / 70.9 to 71.7 dB (eval) from the oracle -- closer -- and two noise-level test bounds derived from the old
// numbers (the ten-step trajectory's perceptual-loss drift, 1.005 % against 1 %; one legacy op's dx, 3.08 % against 3 %) no longer
// held.  Not worth re-deriving bounds in the last round for 0.05 ms; hipcc does not select the mix forms by itself.)
template <class F = FmtA, class FO = F>
__device__ __forceinline__ u32x4 fd_xform8_r(u32x4 raw, f32x4 s0, f32x4 s1, f32x4 h0, f32x4 h1, float slope) {
  typedef f32x2 f32x2_t;
  typedef __attribute__((ext_vector_type(2))) short s16x2_t;
  const f32x2_t sv[4] = {{s0[0], s0[1]}, {s0[2], s0[3]}, {s1[0], s1[1]}, {s1[2], s1[3]}};
  const f32x2_t hv[4] = {{h0[0], h0[1]}, {h0[2], h0[3]}, {h1[0], h1[1]}, {h1[2], h1[3]}};
  u32x4 out;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f32x2_t f = fd_cvt2<F>(raw[i]);
    f = __builtin_elementwise_fma(f, sv[i], hv[i]);
    if (slope == 0.f) {   // uniform: ReLU on the packed result
      const s16x2_t pk = __builtin_bit_cast(s16x2_t, fd_pk2<FO>(f));
      out[i] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(pk, (s16x2_t){0, 0}));
    } else {
      f = __builtin_elementwise_max(f, f * slope);
      out[i] = fd_pk2<FO>(f);
    }
  }
  return out;
}

// prologue transform of one 8-channel unit (no pooling): uniform dispatch on the activation
// F: format of the raw input, FO: format of the result.  <FmtA, FmtG> is what the weight-gradient kernels use: the fp16
// forward input becomes the bf16 operand multiplied with the bf16 dy (always applied there -- with scale 1, shift 0,
// slope 1 it is the plain format conversion).
template <class F = FmtA, class FO = F>
__device__ __forceinline__ u32x4 fd_xform8(u32x4 raw, const float* sc, const float* sh, float slope) {
  if (slope == 0.f) return fd_bn_relu8<F, FO>(raw, sc, sh);
  return fd_pack8<FO>(fd_affine_act<F>(raw, sc, sh, slope));
}

// ---- the row phase of a data gradient on one 8-channel piece (one pixel), shared by the streaming backward kernels ------
//   pre = x * sc + sh;   v = da * (pre > 0 ? w1 : w0);   s1 += v;   s2 += v * x          (BatchNorm's two reductions)
//   o   = ACC == 1 ? o + sc * v  :  ACC == 2 ? sc * v  :  v                              (into / as the gradient of x)
//   act = pre * (pre > 0 ? w1 : w0)        (WANT_ACT: what the forward conv saw, for a fused weight gradient; needs w1 == 1)
// Written on float2 pairs so that hipcc emits v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 (two channels per instruction) and
// two v_cmp + v_cndmask per pair; NO control flow: the element-wise `in ? ... : 0` selects of the first versions were
// compiled into s_and_saveexec / s_cbranch_execz pairs (88 exec-mask regions per two steps of conv1x1_bwdw_kernel).
// Channels past the tensor's last one are the CALLER's business (whole pieces: skip the store; their sums are never read).
template <int ACC, bool WANT_ACT>
__device__ __forceinline__ void fd_row8(const f32x8& da, const f32x8& fx, f32x8& o, const f32x2 (&sc)[4], const f32x2 (&sh)[4], float w1,
                                        float w0, f32x2 (&s1)[4], f32x2 (&s2)[4], f32x8& act) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const f32x2 x2 = {fx[2 * k], fx[2 * k + 1]}, d2 = {da[2 * k], da[2 * k + 1]};
    const f32x2 pre = __builtin_elementwise_fma(x2, sc[k], sh[k]);
    const f32x2 m = {pre[0] > 0.f ? w1 : w0, pre[1] > 0.f ? w1 : w0};
    const f32x2 v = d2 * m;
    s1[k] += v;
    s2[k] = __builtin_elementwise_fma(v, x2, s2[k]);
    f32x2 o2;
    if constexpr (ACC == 1) o2 = __builtin_elementwise_fma(sc[k], v, (f32x2){o[2 * k], o[2 * k + 1]});
    else if constexpr (ACC == 2) o2 = sc[k] * v;
    else o2 = v;
    o[2 * k] = o2[0], o[2 * k + 1] = o2[1];
    if constexpr (WANT_ACT) {
      const f32x2 a2 = pre * m;
      act[2 * k] = a2[0], act[2 * k + 1] = a2[1];
    }
  }
}

// Once per workgroup: BatchNorm -> per-channel (scale, shift) in LDS for channels [0, nch);
// (1, 0) when the conv has no norm.  Workgroup (0,0) also applies the train-mode side
// effects of the norm (running statistics, num_batches_tracked).
__device__ __forceinline__ void fd_fold_bn(const ConvArgs& a, float* sc_lds, float* sh_lds, int nch, int tid, int nt) {
  const bool first = blockIdx.x == 0 && blockIdx.y == 0;
  for (int c = tid; c < nch; c += nt) {
    float sc = 1.f, sh = 0.f;
    if (a.pro_mode == 2) {
      sc = 0.f;
      if (c < a.Cin) {
        const float mean = a.p_mean[c], var = a.p_var[c];
        const float g = a.p_gamma ? a.p_gamma[c] : 1.f, b = a.p_beta ? a.p_beta[c] : 0.f;
        sc = g / sqrtf(var + a.eps);
        sh = b - mean * sc;
        if (a.run_mean != nullptr && first) {
          a.run_mean[c] = (1.f - a.momentum) * a.run_mean[c] + a.momentum * mean;
          a.run_var[c] = (1.f - a.momentum) * a.run_var[c] + a.momentum * var * a.unbias;
        }
      }
    }
    sc_lds[c] = sc;
    sh_lds[c] = sh;
  }
  if (a.pro_mode == 2 && a.nbt != nullptr && first && tid == 0) *a.nbt += 1;
}

// In-kernel finalize of the batch statistics (opt-in: FdStats.mean != NULL; measured SLOWER than the separate launch on
// MI355X in both forms, see fdgan_hip/netplan.py -- kept for hardware where one workgroup's reduction is cheap).  Every
// workgroup has just written its partial row; the last one to arrive (agent-scope counter) reduces all rows in fp64 and writes
// mean / biased variance, saving the fdgan_bn_finalize launch that would otherwise sit between this conv and its consumer.
// Coherence WITHOUT fences: a device-scope release / acquire pair is an L2 write-back + invalidate on MI355X (8 XCDs, 8 L2s)
// -- in every workgroup, right after a kernel that has just written its whole output: measured 2x on the whole forward.
// Instead the partial rows are written and read with RELAXED agent-scope atomic stores / loads (sc1: they go through to the
// memory side and never sit dirty in, or are served stale from, an XCD's L2), the counter is a relaxed agent-scope RMW, and
// the only ordering needed -- a workgroup's row before its increment -- is "the stores have been acknowledged"
// (s_waitcnt vmcnt(0)) ahead of the atomic's issue.  Partial rows: fd_stats_store() below, from the kernels' tails.
// Call from ALL threads of the workgroup after the partial row is stored.  Rows are blockIdx.x-indexed with pitch
// a.stats_cpad.  `scratch`: >= 4.5 KiB of LDS that is free at this point (8-byte aligned).
__device__ __forceinline__ void fd_stats_store(const ConvArgs& a, float* dst, float v) {
  if (a.fin_mean != nullptr)
    __hip_atomic_store(dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else
    *dst = v;
}
__device__ __forceinline__ void fd_finalize_last_block(const ConvArgs& a, int channels, int tid, char* scratch) {
  double(*fd_red)[8][33] = reinterpret_cast<double(*)[8][33]>(scratch);
  volatile unsigned* fd_is_last = reinterpret_cast<volatile unsigned*>(scratch + 2 * 8 * 33 * 8);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this thread's part of the row is at the memory side
  __syncthreads();
  if (tid == 0)
    *fd_is_last = __hip_atomic_fetch_add(a.fin_counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1u : 0u;
  __syncthreads();
  if (!*fd_is_last) return;
  const int rows = (int)gridDim.x;
  for (int c0 = 0; c0 < channels; c0 += 32) {
    const int cl = tid & 31, rg = tid >> 5;   // 32 channels x 8 row groups (first 256 threads)
    double s1 = 0.0, s2 = 0.0;
    if (tid < 256 && c0 + cl < channels)
      for (int r = rg; r < rows; r += 8) {
        float* pr = a.stats + ((long long)r * a.stats_cpad + c0 + cl) * 2;
        s1 += __hip_atomic_load(pr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s2 += __hip_atomic_load(pr + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    if (tid < 256) {
      fd_red[0][rg][cl] = s1;
      fd_red[1][rg][cl] = s2;
    }
    __syncthreads();
    if (tid < 32 && c0 + tid < channels) {
      double t1 = 0.0, t2 = 0.0;
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        t1 += fd_red[0][g][tid];
        t2 += fd_red[1][g][tid];
      }
      const double mean = t1 * a.fin_inv_count;
      double var = t2 * a.fin_inv_count - mean * mean;
      a.fin_mean[c0 + tid] = (float)mean;
      a.fin_var[c0 + tid] = (float)(var < 0.0 ? 0.0 : var);
    }
    __syncthreads();
  }
  if (tid == 0) __hip_atomic_store(a.fin_counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch on this stream
}

// Sum over the 16 lanes of a DPP row (the 16 pixels of an MFMA result row): 4 VALU adds with
// row_ror modifiers, no LDS traffic.  Every lane of the row ends with the full sum.
template <int CTRL>
__device__ __forceinline__ float fd_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float fd_row_sum16(float v) {
  v += fd_dpp<0x128>(v);  // row_ror:8
  v += fd_dpp<0x124>(v);  // row_ror:4
  v += fd_dpp<0x122>(v);  // row_ror:2
  v += fd_dpp<0x121>(v);  // row_ror:1
  return v;
}

// Store 4 consecutive output channels (cout0 .. cout0+3) of one pixel; `off` is the element
// offset of the pixel (n, up*oy, up*ox) in y.  Handles NCHW fp32 / NHWC 16-bit, the 2x2
// replication of the nearest upsample and the ragged last channel group.  F: element format of a 16-bit y.
template <bool ACC = false, class F = FmtA>
__device__ __forceinline__ void fd_store4(const ConvArgs& a, long long off, int cout0, const float (&v)[4]) {
  if (!a.out_nchw_f32) {
    typedef f32x4 f4_t;
    unsigned short* yp = reinterpret_cast<unsigned short*>(a.y) + off + cout0;
    f4_t fv = {v[0], v[1], v[2], v[3]};
    if constexpr (ACC) {   // backward data into a gradient buffer (no upsample, Cout % 4 == 0 checked by the host)
      fv += fd_cvt4<F>(*reinterpret_cast<const u32x2*>(yp));
    }
    const u32x2 bits = fd_pk4<F>(fv);
    if (cout0 + 4 <= a.Cout && !a.upsample) {  // the common case
      *reinterpret_cast<u32x2*>(yp) = bits;
      return;
    }
    const int nrep = a.upsample ? 4 : 1;
    for (int q = 0; q < nrep; ++q) {
      unsigned short* d = yp + (q >> 1) * a.y_sh + (q & 1) * a.y_sw;
      if (cout0 + 4 <= a.Cout) {
        *reinterpret_cast<u32x2*>(d) = bits;
      } else {
        if (cout0 + 0 < a.Cout) d[0] = (unsigned short)(bits[0] & 0xffffu);
        if (cout0 + 1 < a.Cout) d[1] = (unsigned short)(bits[0] >> 16);
        if (cout0 + 2 < a.Cout) d[2] = (unsigned short)(bits[1] & 0xffffu);
      }
    }
  } else {
    float* yp = reinterpret_cast<float*>(a.y) + off + (long long)cout0 * a.y_sc;
    const int nrep = a.upsample ? 4 : 1;
    for (int q = 0; q < nrep; ++q) {
      float* d = yp + (q >> 1) * a.y_sh + (q & 1) * a.y_sw;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (cout0 + r < a.Cout) d[r * a.y_sc] = v[r];
    }
  }
}

// Full-row stores.  An MFMA result lane holds 4 channels (8 bytes) of one pixel, and a wave
// instruction of such stores touches 16 pixels x 32 bytes: measured 2.2 TB/s write-only on
// MI355X against 8 TB/s for 16-byte-per-lane row-contiguous stores (tools/ubench/patterns.hip).
// So a 16-pixel x (CT*16)-channel result tile is transposed through a wave-private LDS
// staging area (row pitch CT*32+16 bytes) and written back as whole rows: CT*2 lanes x 16 B
// per pixel.  `pixoff(q)` returns the element offset of pixel q (0..15) of the tile in y, or a
// negative value for a pixel outside the image.  LDS traffic of one wave is in program order;
// the fences only stop the compiler from reordering the two phases.
template <int CT>
struct RowStore {
  static constexpr int RB = CT * 32;             // bytes of one pixel's channels
  static constexpr int PITCH = RB + 16;
  static constexpr int BYTES = 16 * PITCH;       // staging bytes per wave
  static constexpr int LPP = RB / 16;            // lanes per pixel on the way out
  static constexpr int PPI = 64 / LPP;           // pixels per store instruction (CT = 3: 10, lanes 60..63 idle)
  static constexpr int NIT = (16 + PPI - 1) / PPI;   // store instructions per 16-pixel tile
  static constexpr bool POW2 = (LPP & (LPP - 1)) == 0;
};
template <int CT, bool ACC = false, class FM = FmtA, typename F>
__device__ __forceinline__ void fd_store_row16(const ConvArgs& a, char* tb, const float (&v)[CT][4], int lane,
                                              int cout_base, F pixoff) {
  using R = RowStore<CT>;
  typedef f32x4 f4_t;
  const int m = lane & 15, g = lane >> 4;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    const u32x2 bits = fd_pk4<FM>((f4_t){v[c][0], v[c][1], v[c][2], v[c][3]});
    *reinterpret_cast<u32x2*>(tb + m * R::PITCH + c * 32 + g * 8) = bits;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  const int piece = lane % R::LPP, q0 = lane / R::LPP;
#pragma unroll
  for (int i = 0; i < R::NIT; ++i) {
    const int q = i * R::PPI + q0;
    const bool qok = R::POW2 || (q0 < R::PPI && q < 16);
    u32x4 row = *reinterpret_cast<const u32x4*>(tb + (qok ? q : 0) * R::PITCH + piece * 16);
    const long long off = qok ? pixoff(q) : -1;
    if (off >= 0) {
      u32x4* dst = reinterpret_cast<u32x4*>(reinterpret_cast<unsigned short*>(a.y) + off + cout_base + piece * 8);
      if constexpr (ACC) {   // y += row (the gradient buffer of the forward input), whole 16-byte pieces of a pixel row
        const f32x8 sum = fd_cvt8<FM>(*dst) + fd_cvt8<FM>(row);
        row = fd_pk8<FM>(sum);
      }
      *dst = row;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Same transposition for a caller that already holds the (uniform) address of pixel 0 of the row:
// pixel q lives at yrow + q * y_sw, pixels [0, npix) are stored.  32-bit per-lane offsets only.
template <int CT, class FM = FmtA>
__device__ __forceinline__ void fd_store_row16_ptr(unsigned short* yrow, int y_sw, char* tb, const float (&v)[CT][4],
                                                  int lane, int npix) {
  using R = RowStore<CT>;
  typedef f32x4 f4_t;
  const int m = lane & 15, g = lane >> 4;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    const u32x2 bits = fd_pk4<FM>((f4_t){v[c][0], v[c][1], v[c][2], v[c][3]});
    *reinterpret_cast<u32x2*>(tb + m * R::PITCH + c * 32 + g * 8) = bits;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  const int piece = lane % R::LPP, q0 = lane / R::LPP;
#pragma unroll
  for (int i = 0; i < R::NIT; ++i) {
    const int q = i * R::PPI + q0;
    const bool qok = R::POW2 || (q0 < R::PPI && q < 16);
    const u32x4 row = *reinterpret_cast<const u32x4*>(tb + (qok ? q : 0) * R::PITCH + piece * 16);
    if (qok && q < npix) *reinterpret_cast<u32x4*>(yrow + (unsigned)(q * y_sw + piece * 8)) = row;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int KS, int STRIDE, int POOL, int PT, int CT, int WM, int WN, int TPS, int WD = 0>
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
  static constexpr int W_BYTES = WD ? 0 : TPS * CTB * 1024;   // WD: filter fragments go global -> registers per wave
  static constexpr int IN_UNITS = 4 * NPIXR;
  static constexpr int IN_UPT = (IN_UNITS + NT - 1) / NT;
  static constexpr int W_UNITS = TPS * CTB * 64;
  static constexpr int W_UPT = (W_UNITS + NT - 1) / NT;
  static constexpr int NLOAD = POOL ? 4 : 1;
  static_assert(KK % TPS == 0, "taps per stage must divide KS*KS");
  static_assert(!POOL || (KS == 1 && STRIDE == 1), "pool prologue is for 1x1 convs");
  static unsigned lds_bytes(int nchunk) { return 2 * IN_BYTES + 2 * W_BYTES + nchunk * 32 * 8; }
};

// MK = 1: the backward-data instantiation (masked epilogue, ConvArgs.mk_*); kept out of the forward kernels, whose
// register budget it would double
// WD = 1: "filter direct" main loop for the MFMA-bound shapes (wide 3x3 / 4x4: VGG16, D, refine convs).  The waves of a
// workgroup split the OUTPUT CHANNELS (WN groups of CT*16) and share the pixel tile, so a wave's filter fragments are
// private to it: they go global -> VGPR in fragment order (16 B per lane, lane-linear: the packed image IS the register
// image), one tap ahead, and never touch LDS.  LDS holds only the input halo tile of a 32-channel chunk, which all KS*KS
// taps re-read: ONE barrier per chunk (9 / 16 taps x PT*CT MFMAs per wave) instead of one per tap.
template <int KS, int STRIDE, int POOL, int PT, int CT, int WM, int WN, int TPS, int MK = 0, int WD = 0>
__global__ __launch_bounds__(64 * WM * WN, (WD && PT * CT > 32) ? 1 : 2) void conv_igemm_kernel(ConvArgs a) {
  using C = ConvCfg<KS, STRIDE, POOL, PT, CT, WM, WN, TPS, WD>;
  // element format of x, the filter image and a 16-bit y: forward launches move fp16 activations, the backward-data
  // instantiation moves bf16 gradients (its mk_x, the forward conv's input, is fp16 again)
  using FX = typename FmtSel<MK != 0>::type;
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

  fd_fold_bn(a, sc_lds, sh_lds, a.nchunk * 32, tid, C::NT);

  // ---- per-thread staging maps (fixed across the K loop).  Every unit always loads from a
  // valid address (clamped to the image origin) and is zeroed by a select when it lies outside
  // the image / past Cin: no divergent branch around the loads.
  const unsigned short* xn = a.x + (long long)n * a.x_sn;
  int goff[C::IN_UPT];   // element offset of the unit's first source pixel (0 when clamped)
  int ukg[C::IN_UPT];    // 8-channel group of the unit; -1: outside the image; -2: unit does not exist
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
    ukg[i] = exists ? (inb ? kg : -1) : -2;
    goff[i] = (exists && inb) ? gy * a.x_sh + gx * a.x_sw + kg * 8 : 0;
  }
  const unsigned short* wsrc[C::W_UPT];
  bool wok[C::W_UPT];
#pragma unroll
  for (int i = 0; i < C::W_UPT; ++i) {
    const int j = tid + i * C::NT;
    const int t = j / (C::CTB * 64), rem = j - t * (C::CTB * 64);
    const int tl = rem >> 6, ln = rem & 63;
    const int tile16 = by * C::CTB + tl;
    wok[i] = j < C::W_UNITS && tile16 < a.ntile_total;
    wsrc[i] = wok[i] ? a.w + ((long long)t * a.ntile_total + tile16) * 512 + ln * 8 : a.w;
  }

  u32x4 rin[C::IN_UPT][C::NLOAD];
  u32x4 rw[C::W_UPT];
  const u32x4 zero4 = {0u, 0u, 0u, 0u};

  auto load_in = [&](int chunk) {
#pragma unroll
    for (int i = 0; i < C::IN_UPT; ++i) {
      const bool ok = ukg[i] >= 0 && (chunk * 4 + ukg[i]) < a.Cin8;
      const unsigned short* p = xn + (ok ? goff[i] + chunk * 32 : 0);
      rin[i][0] = *reinterpret_cast<const u32x4*>(p);
      if (POOL) {
        rin[i][1] = *reinterpret_cast<const u32x4*>(p + a.x_sw);
        rin[i][2] = *reinterpret_cast<const u32x4*>(p + a.x_sh);
        rin[i][3] = *reinterpret_cast<const u32x4*>(p + a.x_sh + a.x_sw);
      }
    }
  };
  auto store_in = [&](char* buf, int chunk) {
#pragma unroll
    for (int i = 0; i < C::IN_UPT; ++i) {
      if (ukg[i] == -2) continue;
      const bool ok = ukg[i] >= 0 && (chunk * 4 + ukg[i]) < a.Cin8;
      const int cb = chunk * 32 + (ok ? ukg[i] : 0) * 8;
      u32x4 v;
      if (a.pro_mode == 0 && !POOL) {  // uniform: plain copy
        v = rin[i][0];
      } else if (!POOL) {
        v = fd_xform8<FX>(rin[i][0], sc_lds + cb, sh_lds + cb, a.p_slope);
      } else {
        f32x8 f = fd_affine_act<FX>(rin[i][0], sc_lds + cb, sh_lds + cb, a.p_slope);
        f += fd_affine_act<FX>(rin[i][1], sc_lds + cb, sh_lds + cb, a.p_slope);
        f += fd_affine_act<FX>(rin[i][2], sc_lds + cb, sh_lds + cb, a.p_slope);
        f += fd_affine_act<FX>(rin[i][3], sc_lds + cb, sh_lds + cb, a.p_slope);
        v = fd_pack8<FX>(f * 0.25f);
      }
      lds_write16(buf + (tid + i * C::NT) * 16, ok ? v : zero4);   // zero padding is post-activation
    }
  };
  auto load_w = [&](int chunk, int tg) {
    const long long off = ((long long)chunk * C::KK + tg * TPS) * a.ntile_total * 512;
#pragma unroll
    for (int i = 0; i < C::W_UPT; ++i) rw[i] = *reinterpret_cast<const u32x4*>(wsrc[i] + (wok[i] ? off : 0));
  };
  auto store_w = [&](char* buf) {
#pragma unroll
    for (int i = 0; i < C::W_UPT; ++i) {
      const int j = tid + i * C::NT;
      if (j < C::W_UNITS) lds_write16(buf + j * 16, wok[i] ? rw[i] : zero4);
    }
  };

  f32x4 acc[PT][CT];
#pragma unroll
  for (int p = 0; p < PT; ++p)
#pragma unroll
    for (int c = 0; c < CT; ++c) acc[p][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment base addresses
  const char* xfrag0 = in_lds + kgl * C::PLANE_B + ((wm * PT * STRIDE) * C::IW + m * STRIDE) * 16;
  if constexpr (WD) {
    __syncthreads();  // scale/shift visible
    load_in(0);
    store_in(in_lds, 0);
    // this wave's CT filter fragments of step s = chunk * KK + tap: 1 KiB each, lane-linear.  Wave-uniform base pointer
    // (scalar arithmetic, saddr-form loads) + the lane's 16 bytes.  Output-channel tiles past the filter's last one
    // re-read a valid tile instead of being zeroed: their accumulators are never stored (the epilogue masks by Cout).
    const int wnu = __builtin_amdgcn_readfirstlane(wn);
    int cofs[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      const int tile16 = by * C::CTB + wnu * CT + c;
      cofs[c] = (tile16 < a.ntile_total ? tile16 : 0) * 512;
    }
    // Tap order is COLUMN-major (dx outer, dy inner): for one dx the PT + KS - 1 input-row fragments are read from LDS
    // once and serve all KS row taps (output row p at tap dy reads input row p + dy) -- 30 fragment reads per chunk
    // instead of 72 for a 3x3 on 8 rows.  Filter fragments are requested TWO taps ahead in that order (an L2 round trip
    // under load is longer than one tap's MFMAs); the sched_barrier keeps the compiler from sinking the request down to
    // its first use.
    static_assert(STRIDE == 1 && !POOL, "filter-direct kernels are stride-1, unpooled");
    const long long wstep = (long long)a.ntile_total * 512;             // elements per (chunk, tap)
    const unsigned short* wch = a.w + lane * 8;                          // this chunk's taps; the lane's 16 bytes
    auto wload = [&](u32x4 (&dst)[CT], const unsigned short* base, int e) {   // e: position in execution order
      const int tap = (e % KS) * KS + (e / KS);                          // e = dx * KS + dy  ->  tap = dy * KS + dx
#pragma unroll
      for (int c = 0; c < CT; ++c) dst[c] = *reinterpret_cast<const u32x4*>(base + tap * wstep + cofs[c]);
    };
    u32x4 wcur[CT], wnx1[CT], wnx2[CT];
    wload(wcur, wch, 0);
    wload(wnx1, wch, 1);
    __syncthreads();
    // s_setprio around the MFMA blocks: +3..9 % (the co-resident wave's loads / LDS reads yield to the MFMA issue);
    // VAR bit 0 (tuning): store the next chunk's input tile early instead of before the barrier (mixed: off)
    constexpr int VAR = (WD >> 1) ^ 2;
    for (int chunk = 0; chunk < a.nchunk; ++chunk) {
      const bool has_next = (chunk + 1) < a.nchunk;
      if (has_next) load_in(chunk + 1);
      const unsigned short* wnext = wch + (has_next ? (long long)C::KK * wstep : 0);   // past the end: re-read (unused)
      const char* xb = xfrag0 + (chunk & 1) * C::IN_BYTES;
#pragma unroll
      for (int dx = 0; dx < KS; ++dx) {
        u32x4 xr[PT + KS - 1];
#pragma unroll
        for (int r = 0; r < PT + KS - 1; ++r) xr[r] = lds_read16(xb + (r * C::IW + dx) * 16);
#pragma unroll
        for (int dy = 0; dy < KS; ++dy) {
          const int e = dx * KS + dy;
          if (e + 2 < C::KK) wload(wnx2, wch, e + 2);
          else wload(wnx2, wnext, e + 2 - C::KK);
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (VAR & 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll
          for (int p = 0; p < PT; ++p)
#pragma unroll
            for (int c = 0; c < CT; ++c)
              acc[p][c] = fd_mfma<FX>(wcur[c], xr[p + dy], acc[p][c]);
          if constexpr (VAR & 2) __builtin_amdgcn_s_setprio(0);
#pragma unroll
          for (int c = 0; c < CT; ++c) {
            wcur[c] = wnx1[c];
            wnx1[c] = wnx2[c];
          }
        }
        // the other input buffer is free for the whole chunk (every wave left it at the previous barrier): fill it while
        // the MFMA pipe is busy with the remaining taps instead of in the bubble before the barrier
        if constexpr (VAR & 1) {
          if (dx == KS - 2 && has_next) store_in(in_lds + ((chunk + 1) & 1) * C::IN_BYTES, chunk + 1);
        }
      }
      wch = wnext;
      if constexpr (!(VAR & 1)) {
        if (has_next) store_in(in_lds + ((chunk + 1) & 1) * C::IN_BYTES, chunk + 1);
      }
      __syncthreads();
    }
  } else {
  const int nsteps = a.nchunk * C::NSPC;
  __syncthreads();  // scale/shift visible
  load_in(0);
  load_w(0, 0);
  store_in(in_lds, 0);
  store_w(w_lds);
  __syncthreads();

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
      u32x4 wf[CT], xf[PT];
#pragma unroll
      for (int c = 0; c < CT; ++c)
        wf[c] = lds_read16(wb + (t * C::CTB + c) * 1024);
#pragma unroll
      for (int p = 0; p < PT; ++p)
        xf[p] = lds_read16(xb + ((p * STRIDE + dy) * C::IW + dx) * 16);
#pragma unroll
      for (int p = 0; p < PT; ++p)
#pragma unroll
        for (int c = 0; c < CT; ++c)
          acc[p][c] = fd_mfma<FX>(wf[c], xf[p], acc[p][c]);
    }

    if (has_next) store_w(w_lds + ((s + 1) & 1) * C::W_BYTES);
    if (new_chunk) store_in(in_lds + (chunk1 & 1) * C::IN_BYTES, chunk1);
    __syncthreads();
  }
  }

  // ---- epilogue: bias, activation, store, batch statistics
  const int col = ox0 + m;
  const int up = a.upsample ? 2 : 1;
  // staging for the row stores: one RowStore<CT> area per wave at the start of LDS (the K loop
  // ended with a barrier, so the input ring is free); statistics scratch behind it
  char* tb = smem + wave * RowStore<CT>::BYTES;
  float* red = reinterpret_cast<float*>(smem + WM * WN * RowStore<CT>::BYTES);  // [waves][CT*16][2]
  const int cbase = by * C::BN + wn * CT * 16;
  const bool rowstore = a.y_vec16 && cbase + CT * 16 <= a.Cout;   // uniform
  float bv[CT][4];
#pragma unroll
  for (int c = 0; c < CT; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = cbase + c * 16 + kgl * 4 + r;
      // (unconditional load of a clamped index: inside `cond ? a.bias[co] : 0` each of the CT x 4 loads is waited for before the next is
      // issued -- eight dependent L2 round trips at the tail of every workgroup of VGG16's convs)
      const bool bok = a.bias != nullptr && co < a.CoutW;
      const float bval = (a.bias != nullptr ? a.bias : reinterpret_cast<const float*>(a.w))[bok ? co : 0];
      bv[c][r] = bok ? bval : 0.f;
    }
  if constexpr (MK != 0) {
    // Backward data.  Row phase only: each 16-pixel x (CT*16)-channel accumulator tile goes through the wave's LDS staging
    // area (as the forward's row stores do) and comes back as 16-byte pieces of pixel rows; the forward input x and (in
    // accumulate mode) the gradient buffer are read in that same shape -- whole coalesced rows, one latency for both --
    // and the lane keeps the 8 channels it owns for the whole tile: 16 coefficient registers, 16 running sums.
    // v = da * act'(bn(x));  sums (v, v * x) per channel;  store v, or y += gamma * rstd * v.
    // (Round 4, measured and taken out again: the x / G loads of the row loop below sit inside `if (ok ...)` and are each waited
    // for at the join behind them.  Issued unconditionally from clamped addresses -- with or without a second register set one
    // accumulator row ahead -- the kernels need 12-16 more registers, lose a wave per SIMD and get SLOWER: D's 4x4 data gradient
    // 587 -> 674 us, VGG16's 3x3 137 -> 149 us, the training step 27.4 -> 28.9 ms.  Occupancy, not this tail, is what they live on.)
    using R = RowStore<CT>;
    typedef f32x4 f4_t;
    float* msc = red + WM * WN * CT * 16 * 2;   // [BN] scale, [BN] shift of this workgroup's channels
    float* msh = msc + C::BN;
    for (int cl = tid; cl < C::BN; cl += C::NT) {
      const int co = by * C::BN + cl;
      float sc = 1.f, sh = 0.f;
      if (a.mk_mode == 2 && co < a.Cout) {
        const float gm = a.mk_gamma ? a.mk_gamma[co] : 1.f, bt = a.mk_beta ? a.mk_beta[co] : 0.f;
        sc = gm / sqrtf(a.mk_var[co] + a.mk_eps);
        sh = bt - a.mk_mean[co] * sc;
      }
      msc[cl] = sc;
      msh[cl] = sh;
    }
    __syncthreads();
    const int piece = lane % R::LPP, q0 = lane / R::LPP;
    const int cg = cbase + piece * 8;             // first of this lane's 8 channels
    const bool ch_ok = cg < a.Cout;               // host: pixel pitch >= Cout rounded up to 8
    float sc8[8], sh8[8], s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      sc8[e] = msc[wn * CT * 16 + piece * 8 + e];
      sh8[e] = msh[wn * CT * 16 + piece * 8 + e];
      s1[e] = s2[e] = 0.f;
    }
    unsigned short* ybase = reinterpret_cast<unsigned short*>(a.y);
#pragma unroll
    for (int p = 0; p < PT; ++p) {
      const int row = oy0 + wm * PT + p;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        const u32x2 bits = fd_pk4<FmtG>((f4_t){acc[p][c][0], acc[p][c][1], acc[p][c][2], acc[p][c][3]});
        *reinterpret_cast<u32x2*>(tb + m * R::PITCH + c * 32 + kgl * 8) = bits;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      u32x4 dav[R::NIT], xv[R::NIT], gv[R::NIT];
      long long offs[R::NIT];
#pragma unroll
      for (int i = 0; i < R::NIT; ++i) {
        const int q = i * R::PPI + q0;
        const bool qok = R::POW2 || (q0 < R::PPI && q < 16);
        const bool ok = qok && ch_ok && row < a.Ho && ox0 + q < a.Wo;
        offs[i] = ok ? (long long)n * a.y_sn + (long long)row * a.y_sh + (long long)(ox0 + q) * a.y_sw + cg : -1;
        dav[i] = *reinterpret_cast<const u32x4*>(tb + (qok ? q : 0) * R::PITCH + piece * 16);
        xv[i] = gv[i] = u32x4{0u, 0u, 0u, 0u};
        if (ok && a.mk_mode != 0) {
          xv[i] = *reinterpret_cast<const u32x4*>(a.mk_x + (long long)n * a.mk_sn + (long long)row * a.mk_sh + (long long)(ox0 + q) * a.mk_sw + cg);
        }
        if (ok && a.mk_acc == 1) gv[i] = *reinterpret_cast<const u32x4*>(ybase + offs[i]);
      }
#pragma unroll
      for (int i = 0; i < R::NIT; ++i) {
        if (offs[i] < 0) continue;
        const f32x8 da = fd_cvt8<FmtG>(dav[i]);
        const f32x8 fx = fd_cvt8<FmtA>(xv[i]);      // the forward conv's input (fp16)
        f32x8 o = fd_cvt8<FmtG>(gv[i]);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float pre = fmaf(fx[e], sc8[e], sh8[e]);
          const float v = (cg + e < a.Cout) ? da[e] * ((pre > 0.f || a.mk_mode == 0) ? 1.f : a.mk_slope) : 0.f;
          s1[e] += v;
          s2[e] += v * fx[e];
          o[e] = a.mk_acc ? fmaf(sc8[e], v, o[e]) : v;   // o = 0 unless accumulating
        }
        *reinterpret_cast<u32x4*>(ybase + offs[i]) = fd_pk8<FmtG>(o);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if (a.stats != nullptr) {   // lanes LPP apart own the same channels
      if constexpr (R::POW2) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
#pragma unroll
          for (int dlt = R::LPP; dlt < 64; dlt <<= 1) {
            s1[e] += __shfl_xor(s1[e], dlt, 64);
            s2[e] += __shfl_xor(s2[e], dlt, 64);
          }
        }
        if (q0 == 0)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int idx = (wave * CT * 16 + piece * 8 + e) * 2;
            red[idx] = s1[e];
            red[idx + 1] = s2[e];
          }
      } else {   // 6 lanes per pixel: lanes piece + LPP k (k < PPI) own the same channels, which is not a butterfly.  Summed in a
                 // FIXED order through ds_bpermute, once per kernel (LDS atomics here made D's BatchNorm gradients differ in the
                 // last bit from run to run: tests/test_hip_models.py::test_training_step_full_size_configs2)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float t1 = 0.f, t2 = 0.f;
#pragma unroll
          for (int k = 0; k < R::PPI; ++k) {
            t1 += __shfl(s1[e], piece + R::LPP * k, 64);
            t2 += __shfl(s2[e], piece + R::LPP * k, 64);
          }
          s1[e] = t1, s2[e] = t2;
        }
        if (q0 == 0)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int idx = (wave * CT * 16 + piece * 8 + e) * 2;
            red[idx] = s1[e];
            red[idx + 1] = s2[e];
          }
      }
    }
  } else {
#pragma unroll
  for (int p = 0; p < PT; ++p) {
    const int row = oy0 + wm * PT + p;
    float v[CT][4];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float t = acc[p][c][r] + bv[c][r];
        v[c][r] = fmaxf(t, a.e_slope * t);
      }
    auto pixoff = [&](int q) -> long long {
      return (row < a.Ho && ox0 + q < a.Wo) ? (long long)n * a.y_sn + (long long)row * a.y_sh + (long long)(ox0 + q) * a.y_sw : -1;
    };
    if (rowstore) {
      fd_store_row16<CT, false, FX>(a, tb, v, lane, cbase, pixoff);
    } else if (row < a.Ho && col < a.Wo) {
      const long long off = (long long)n * a.y_sn + (long long)(up * row) * a.y_sh + (long long)(up * col) * a.y_sw;
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        const int cout0 = cbase + c * 16 + kgl * 4;
        if (cout0 < a.Cout) fd_store4<false, FX>(a, off, cout0, v[c]);
      }
    }
  }
  }   // MK == 0: the forward epilogue
  if (a.stats != nullptr && MK == 0) {
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int p = 0; p < PT; ++p) {
        const bool valid = (oy0 + wm * PT + p) < a.Ho && col < a.Wo;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float t = acc[p][c][r] + bv[c][r];
          const float v = fmaxf(t, a.e_slope * t);
          s1[r] += valid ? v : 0.f;
          s2[r] += valid ? v * v : 0.f;
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s1[r] = fd_row_sum16(s1[r]);
        s2[r] = fd_row_sum16(s2[r]);
      }
      if (m == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int idx = (wave * CT * 16 + c * 16 + kgl * 4 + r) * 2;
          red[idx] = s1[r];
          red[idx + 1] = s2[r];
        }
      }
    }
  }

  if (a.stats != nullptr) {
    __syncthreads();
    for (int cl = tid; cl < C::BN; cl += C::NT) {
      const int wn_ = cl / (CT * 16), idx = cl - wn_ * (CT * 16);
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int w_ = 0; w_ < WM; ++w_) {
        t1 += red[((w_ * WN + wn_) * CT * 16 + idx) * 2];
        t2 += red[((w_ * WN + wn_) * CT * 16 + idx) * 2 + 1];
      }
      float* dst = a.stats + ((long long)blockIdx.x * a.stats_cpad + by * C::BN + cl) * 2;
      dst[0] = t1;
      dst[1] = t2;
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
#define FD_CONV_DISPATCH(KS_, ST_, POOL_, PT_, CT_, WM_, WN_, TPS_, NAME_) FD_CONV_DISPATCH_X(KS_, ST_, POOL_, PT_, CT_, WM_, WN_, TPS_, 0, NAME_)
#define FD_CONV_DISPATCH_X(KS_, ST_, POOL_, PT_, CT_, WM_, WN_, TPS_, MK_, NAME_) FD_CONV_DISPATCH_W(KS_, ST_, POOL_, PT_, CT_, WM_, WN_, TPS_, MK_, 0, NAME_)
#define FD_CONV_DISPATCH_W(KS_, ST_, POOL_, PT_, CT_, WM_, WN_, TPS_, MK_, WD_, NAME_)                          \
  do {                                                                                                          \
    using C = ConvCfg<KS_, ST_, POOL_, PT_, CT_, WM_, WN_, TPS_, WD_>;                                                \
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
    auto kfn = &conv_igemm_kernel<KS_, ST_, POOL_, PT_, CT_, WM_, WN_, TPS_, MK_, WD_>;                              \
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

// one translation unit per kernel family (parallel compilation)
int conv_dispatch_k1(ConvArgs& a, long long nimg, int cout_total, int stride, bool pool, FdConvInfo* info,
                     long long stats_cap, bool dry, hipStream_t stream);
bool conv1x1_xs_fits(int cout_total, int cin);
int conv_dispatch_k1_xs(ConvArgs& a, long long nimg, int cout_total, bool pool, FdConvInfo* info,
                        long long stats_cap, bool dry, hipStream_t stream);
bool conv1x1_ds_fits(const ConvArgs& a, int cout_total, bool pool, int w_layout);
int conv_dispatch_k1_ds(ConvArgs& a, FdConvInfo* info, long long stats_cap, bool dry, hipStream_t stream);
int conv_dispatch_k3(ConvArgs& a, long long nimg, int cout_total, int stride, bool pool, FdConvInfo* info,
                     long long stats_cap, bool dry, hipStream_t stream);
bool conv3x3_pw_fits(int cout_total, int cin);
bool conv3x3_rs_fits(const ConvArgs& a, int cout_total);
// weight gradient of the growth conv, rows staged in their memory layout and read with ds_read_b64_tr_b16 (conv_wgrad_tr.hip)
struct WgradRowsArgs {
  const unsigned short* x;       // raw forward input (NHWC fp16 view)
  long long x_sn;
  int x_sh, x_sw;
  const unsigned short* dy;      // gradient of the conv output (NHWC bf16 view, 32 channels)
  long long dy_sn;
  int dy_sh, dy_sw;
  int H, W, Cin, Cout, Ho, Wo, pad;
  int xblocks, seg_rows, segs;   // column blocks per image; output rows per work item; row segments per (image, block)
  int pro_mode;
  float p_slope, eps;
  const float *p_mean, *p_var, *p_gamma, *p_beta;
  float* part;                   // per-item partial sums in accumulator order (conv_wgrad_tr.hip)
  float* bias_part;              // [items][cout tiles][32] per-item sums of dy (bias gradient) or NULL
  int dbg_skip;                  // FDGAN_DEBUG_PHASES (tuning builds, results wrong): 1 no partial stores (every row-walking kernel);
                                 // conv_wgrad_r3 / r4 also take 8 / 32 / 256 (see there).  Bits 2 and 4 (no MFMA loop / no staging) are GONE
                                 // from conv_wgrad_tr since its row loop lost every branch between a load and its store (round 5)
  int vgx, vgy, vgz;             // conv_wgrad_r4: the logical (cin tile, item, cout slice) grid behind its 1-D XCD-aware launch
  int zt;                        // conv_wgrad_tr: 32-filter groups x filter-row groups of the partial-sum layout (>= gridDim.z)
};
// streaming 1x1 data gradient + prologue backward (conv1x1_bwd.hip)
bool conv1x1_bwd_fits(const FdTensor* dy, const FdTensor* fwd_x, const FdTensor* dpre);
int conv1x1_bwd_launch(const FdTensor* dy, const void* w_packed, const FdTensor* fwd_x, const FdPrologue* pro, const FdTensor* dpre,
                       int accumulate, float* partial, long long capacity_floats, long long* rows_out, long long* cpad_out,
                       hipStream_t stream, float* wpart = nullptr, long long wpart_floats = 0, long long* wsplit_out = nullptr,
                       const FdTensor* dy_affine_x = nullptr, const float* dy_affine_b = nullptr, const float* dy_affine_c = nullptr);
// [nsplit][numel] partial weight gradients -> out (+= when accumulate), fixed summation order (conv_bwd.hip)
int fd_wgrad_reduce(const float* part, float* out, long long numel, int nsplit, int accumulate, hipStream_t stream);
// few outputs x many partial rows (bias gradients); pitch: floats between rows
int fd_wgrad_reduce_wide(const float* part, float* out, long long numel, long long pitch, int nsplit, int accumulate, hipStream_t stream);
// row-streaming 3x3 data gradient of the growth conv + prologue backward (conv3x3_bwd.hip)
bool conv3x3_bwd_fits(const FdTensor* dy, const FdTensor* fwd_x, const FdTensor* dpre, const FdConvDesc* d);
int conv3x3_bwd_launch(const FdTensor* dy, const void* w_packed, const FdTensor* fwd_x, const FdPrologue* pro, const FdTensor* dpre,
                       int accumulate, float* partial, long long capacity_floats, long long* rows_out, long long* cpad_out,
                       hipStream_t stream);
// 1x1 weight gradient with transpose reads (conv_wgrad1x1_tr.hip)
bool conv_wgrad1x1_tr_fits(const FdTensor* x, const FdTensor* dy, int cout, int ksize, int stride, bool pool, bool has_bias);
int conv_wgrad1x1_tr_launch(const FdTensor* x, const FdTensor* dy, int pro_mode, float p_slope, float eps, const float* mean,
                            const float* var, const float* gamma, const float* beta, float* workspace, long long workspace_floats,
                            long long* nsplit_out, hipStream_t stream);
// few-channel weight gradient as one GEMM over all taps (conv_wgrad_small.hip)
int conv_wgrad_small_launch(const FdTensor* x, const FdTensor* dy, int cout, int ksize, int stride, int pad, int pro_mode, float p_slope,
                            float eps, const float* mean, const float* var, const float* gamma, const float* beta, bool want_bias,
                            float* workspace, long long workspace_floats, long long* nsplit_out, hipStream_t stream);
int conv_wgrad_tr_variant(int cout, int cin, int ksize, int stride, int pad, bool pool);
int conv_wgrad_tr_launch(int variant, WgradRowsArgs& a, long long nimg, float* workspace, long long workspace_floats, float* dw,
                         float* dbias, int accumulate, hipStream_t stream, FdTrReduceJob* job = nullptr, int defer = 0);
// job: filled with the description of the reduction over the kernel's partial sums; defer != 0 (and no dbias): that reduction is
// NOT launched -- the caller runs a table of them later (conv_wgrad_tr_reduce_batch; jobs_device: the table in device memory)
int conv_wgrad_tr_reduce_batch(const FdTrReduceJob* jobs_device, long long njobs, long long total_groups, hipStream_t stream);

int conv_dispatch_k3_rs(ConvArgs& a, long long nimg, int cout_total, FdConvInfo* info, long long stats_cap, bool dry,
                        hipStream_t stream);
int conv_dispatch_k3_pw(ConvArgs& a, long long nimg, int cout_total, FdConvInfo* info, long long stats_cap, bool dry,
                        hipStream_t stream);
int conv_dispatch_k4(ConvArgs& a, long long nimg, int cout_total, int stride, bool pool, FdConvInfo* info,
                     long long stats_cap, bool dry, hipStream_t stream);
// the one-filter convolution that ends the discriminators, and its data gradient (conv_c1.hip)
bool conv_cout1_fits(const ConvArgs& a, int cout_total, int ksize, int stride, bool pool);
int conv_cout1_launch(const ConvArgs& a, long long nimg, int ksize, FdConvInfo* info, bool dry, hipStream_t stream);
int wgrad_cout1_launch(const FdTensor* x, const FdTensor* dy, int ksize, int stride, int pad, int pro_mode, float slope, float eps,
                       const float* mean, const float* var, const float* gamma, const float* beta, float* workspace,
                       long long workspace_floats, long long* nsplit_out, hipStream_t stream);
int dgrad_cout1_launch(const FdTensor* dy, const void* w_packed_flipped, const FdTensor* fwd_x, const FdPrologue* fwd_pro, const FdTensor* dpre,
                       int accumulate, const FdConvDesc* d, hipStream_t stream);
// the image-reading first layers (3 -> 64 3x3, 9 -> 36 4x4 stride 2): k = (tap, channel) with the channels padded to 4 / 16 (conv_sc.hip)
int conv_sc_variant(const ConvArgs& a, int cout_total, int ksize, int stride, bool pool);
int conv_sc_launch(int variant, ConvArgs& a, long long nimg, FdConvInfo* info, long long stats_cap, bool dry, hipStream_t stream);
extern unsigned long long* g_fd_debug_timing;
// tanh / sigmoid applied in place on what a conv stored (elementwise.hip)
int fd_act_inplace(const FdTensor* y, int act, hipStream_t stream);
