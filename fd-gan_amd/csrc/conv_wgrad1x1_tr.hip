// conv_wgrad1x1_tr.hip -- weight gradient of the dense-layer bottleneck (1x1, 128 filters, any Cin) with the LDS
// transpose read:   dW[co][ci] = sum_p dy[p][co] * a[p][ci],   a = relu(bn(x)) recomputed from the raw input.
//
// The per-tap kernel of conv_bwd.hip transposes both operands in registers (v_perm_b32 + ds_write_b64): on the
// 224 -> 128 layer at 256x256 its compute side (transposition, LDS, MFMA) takes 138 us next to 173 us of loads, and
// the two only partly overlap (248 us).  Here both operands are staged in their memory layout -- x through the
// prologue transform and one ds_write_b128, dy by a plain copy -- and every MFMA fragment is two ds_read_b64_tr_b16
// (see conv_wgrad_tr.hip for the instruction); the rows of step s+1 are in flight while step s computes, LDS is double
// buffered, one barrier per 64-pixel step.
// Workgroup = 8 waves, 128 cout x 128 cin of one pixel split: wave (w & 1, w >> 1) owns 4 cout tiles x 2 cin tiles
// (4 + 2 fragment reads per 8 MFMAs).  Pixels are walked in flattened N*H*W order (dense views).  The result leaves
// through LDS as whole [co][128 ci] rows, so the split partials have the layout wgrad_reduce expects.
// Reference: autograd of conv1 of torchvision's _DenseLayer as used by /root/reference/models/dehaze1113.py:713-724.
#include <stdlib.h>

#include "conv_igemm.h"

namespace {

constexpr int W1_PX = 64;                 // pixels per step
constexpr int W1_ROW = 256;               // bytes per staged pixel of either operand (128 channels)
constexpr int W1_BUF = 2 * W1_PX * W1_ROW;   // x tile + dy tile of one step (32 KB)
#define W1_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

typedef short s16x4 __attribute__((ext_vector_type(4)));

struct Wg1Args {
  const unsigned short* x;    // [P][x_pitch]
  int x_pitch, Cin;
  const unsigned short* dy;   // [P][dy_pitch], 128 channels
  int dy_pitch;
  long long P, split_px;      // pixels per split: a multiple of 64
  int nsplit, tiles_ci;
  int pro_mode;
  float p_slope, eps;
  const float *p_mean, *p_var, *p_gamma, *p_beta;
  float* part;                // [nsplit][128][Cin]
};

__device__ __forceinline__ bf16x8 w1_frag(const char* p0, const char* p1) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p1));
  return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}
// byte offset of the 32-byte (16-channel) group c16 of staged pixel pix: the eight pixels a half-wave touches
// (p..p+3, p+8..p+11) land in eight different bank groups (conv_wgrad_tr.hip)
__device__ __forceinline__ int w1_off(int pix, int c16) {
  return pix * W1_ROW + ((c16 ^ ((pix & 3) | (((pix >> 3) & 1) << 2))) << 5);
}

__global__ __launch_bounds__(512) void conv_wgrad1x1_tr_kernel(Wg1Args a) {
  extern __shared__ __attribute__((aligned(16))) char w1_lds[];
  float* sc_s = reinterpret_cast<float*>(w1_lds + 2 * W1_BUF);
  float* sh_s = sc_s + 128;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // XCD-aware numbering: the cin slices of one pixel split read the same dy pixels -- keep them on one XCD, back to back
  int item = blockIdx.x;
  {
    const int per_xcd = gridDim.x >> 3;
    if (item < per_xcd * 8) item = (item & 7) * per_xcd + (item >> 3);
  }
  const int tci = item % a.tiles_ci, split = item / a.tiles_ci;
  const int ci0 = tci * 128;
  if (tid < 128) {
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
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  // staging: 1024 sixteen-byte units per operand and step, two per thread: (pixel, chunk) = (u / 16, u % 16)
  const int chunk = tid & 15, spix = tid >> 4;           // + 32 pixels for the second unit
  const bool x_ok = ci0 + chunk * 8 < a.Cin;
  const long long p_begin = (long long)split * a.split_px;
  const long long p_end = p_begin + a.split_px < a.P ? p_begin + a.split_px : a.P;
  const unsigned short* xsrc = a.x + ci0 + chunk * 8;
  const unsigned short* dsrc = a.dy + chunk * 8;
  int dst[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) dst[k] = w1_off(spix + 32 * k, chunk >> 1) + ((chunk & 1) << 4);
  u32x4 xr[2], dr[2];
  auto load_step = [&](long long p0) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const long long p = p0 + spix + 32 * k;
      xr[k] = dr[k] = zero4;
      if (p < p_end) {
        if (x_ok) xr[k] = *reinterpret_cast<const u32x4*>(xsrc + p * a.x_pitch);
        dr[k] = *reinterpret_cast<const u32x4*>(dsrc + p * a.dy_pitch);
      }
    }
  };
  auto store_step = [&](char* buf, long long p0) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      u32x4 v = xr[k];
      // fp16 forward input -> the bf16 operand (always: without a prologue scale 1, shift 0, slope 1 = the conversion)
      if (x_ok && p0 + spix + 32 * k < p_end) v = fd_xform8<FmtA, FmtG>(v, sc_s + chunk * 8, sh_s + chunk * 8, a.pro_mode != 0 ? a.p_slope : 1.f);
      lds_write16(buf + dst[k], v);                        // pixels past the end stay zero: they add nothing
      lds_write16(buf + W1_PX * W1_ROW + dst[k], dr[k]);
    }
  };
  // fragments: lane (g, i): k rows 8 g + (i >> 2) (+ 4 for the second read), 4-channel piece i & 3
  const int g = lane >> 4, i = lane & 15;
  const int kpix = 8 * g + (i >> 2), piece = (i & 3) * 8;
  const int wco = (wave & 1) * 4, wci = (wave >> 1) * 2;   // first cout / cin tile of this wave
  f32x4 acc[4][2];
#pragma unroll
  for (int c = 0; c < 4; ++c) acc[c][0] = acc[c][1] = f32x4{0.f, 0.f, 0.f, 0.f};

  __syncthreads();                                         // sc_s / sh_s
  if (p_begin < p_end) {
    load_step(p_begin);
    store_step(w1_lds, p_begin);
  }
  int cur = 0;
  for (long long p0 = p_begin; p0 < p_end; p0 += W1_PX) {
    const bool more = p0 + W1_PX < p_end;
    if (more) load_step(p0 + W1_PX);                       // in flight during this step's MFMAs
    W1_BARRIER();                                          // this step's tiles written; the other buffer's readers done
    const char* xt = w1_lds + cur * W1_BUF;
    const char* dt = xt + W1_PX * W1_ROW;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      const int pa = 32 * sub + kpix;
      bf16x8 af[4], bf[2];
#pragma unroll
      for (int c = 0; c < 4; ++c) af[c] = w1_frag(dt + w1_off(pa, wco + c) + piece, dt + w1_off(pa + 4, wco + c) + piece);
#pragma unroll
      for (int j = 0; j < 2; ++j) bf[j] = w1_frag(xt + w1_off(pa, wci + j) + piece, xt + w1_off(pa + 4, wci + j) + piece);
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[c][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[c], bf[j], acc[c][j], 0, 0, 0);
    }
    cur ^= 1;
    if (more) store_step(w1_lds + cur * W1_BUF, p0 + W1_PX);   // the buffer read one barrier ago
  }
  // ---- result: D layout column (lane & 15) = cin, rows (lane >> 4) * 4 + r = cout -> LDS [128 co][128 ci] fp32 -> rows
  W1_BARRIER();
  float* out = reinterpret_cast<float*>(w1_lds);           // 64 KB = both step buffers
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) out[((wco + c) * 16 + g * 4 + r) * 128 + (wci + j) * 16 + i] = acc[c][j][r];
  __syncthreads();
  float* dwp = a.part + (long long)split * 128 * a.Cin;
  for (int u = tid; u < 128 * 32; u += 512) {              // 16-byte pieces of the [128][128] tile
    const int co = u >> 5, c4 = (u & 31) * 4;
    const f32x4 v = *reinterpret_cast<const f32x4*>(out + co * 128 + c4);
    float* d = dwp + (long long)co * a.Cin + ci0 + c4;
    if (ci0 + c4 + 4 <= a.Cin && (a.Cin & 3) == 0) {
      *reinterpret_cast<f32x4*>(d) = v;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (ci0 + c4 + e < a.Cin) d[e] = v[e];
    }
  }
}

}  // namespace

bool conv_wgrad1x1_tr_fits(const FdTensor* x, const FdTensor* dy, int cout, int ksize, int stride, bool pool, bool has_bias) {
  auto dense = [](const FdTensor* t) {
    return t->stride[3] == 1 && t->stride[1] == t->w * t->stride[2] && t->stride[0] == t->h * t->stride[1] && t->stride[2] % 8 == 0 &&
           ((uintptr_t)t->ptr & 15) == 0;
  };
  const long long P = x->n * x->h * x->w;
  return ksize == 1 && stride == 1 && !pool && !has_bias && cout == 128 && dy->c == 128 && x->c >= 64 && dense(x) && dense(dy) &&
         P % W1_PX == 0 && x->stride[2] >= (x->c + 7) / 8 * 8 && FD_TUNE_GETENV("FDGAN_DEBUG_NO_WGRAD1X1_TR") == nullptr;
}

/* Partials [nsplit][128][Cin] into `workspace`; returns nsplit through *nsplit_out (the caller runs wgrad_reduce). */
int conv_wgrad1x1_tr_launch(const FdTensor* x, const FdTensor* dy, int pro_mode, float p_slope, float eps, const float* mean,
                            const float* var, const float* gamma, const float* beta, float* workspace, long long workspace_floats,
                            long long* nsplit_out, hipStream_t stream) {
  Wg1Args a{};
  a.x = static_cast<const unsigned short*>(x->ptr), a.x_pitch = (int)x->stride[2], a.Cin = (int)x->c;
  a.dy = static_cast<const unsigned short*>(dy->ptr), a.dy_pitch = (int)dy->stride[2];
  a.P = x->n * x->h * x->w;
  a.tiles_ci = (a.Cin + 127) / 128;
  a.pro_mode = pro_mode, a.p_slope = p_slope, a.eps = eps;
  a.p_mean = mean, a.p_var = var, a.p_gamma = gamma, a.p_beta = beta;
  const long long numel = 128LL * a.Cin;
  long long nsplit = 512 / a.tiles_ci;                     // two resident workgroups per CU
  const long long max_by_px = (a.P + 1023) / 1024;
  if (nsplit > max_by_px) nsplit = max_by_px;
  if (nsplit * numel > workspace_floats) nsplit = workspace_floats / numel;
  if (nsplit < 1) return 1;                                // caller falls back to the per-tap kernel
  a.nsplit = (int)nsplit;
  a.split_px = ((a.P + nsplit - 1) / nsplit + W1_PX - 1) / W1_PX * W1_PX;
  a.part = workspace;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad1x1_tr_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) FD_FAIL(FD_ELAUNCH, "hipFuncSetAttribute(conv_wgrad1x1_tr): %s", hipGetErrorString(e));
    attr_done = true;
  }
  *nsplit_out = nsplit;
  return fd_launch(&conv_wgrad1x1_tr_kernel, "conv_wgrad1x1_tr", dim3((unsigned)(a.tiles_ci * nsplit)), dim3(512), 2 * W1_BUF + 1024, a, stream);
}
