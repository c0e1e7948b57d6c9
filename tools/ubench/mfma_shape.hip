// mfma_shape.hip -- v_mfma_f32_16x16x32_f16 vs v_mfma_f32_32x32x16_f16 INSIDE the filter-direct loop structure of
// csrc/conv_igemm.h (WD = 1): VERDICT r4 next #1(c) asks for the 32x32x16 core to be measured instead of argued on paper.
//
// Both variants are the same 3x3 stride-1 implicit-GEMM skeleton as conv_igemm_kernel<3,1,0,8,2,1,4,9,0,1> ("conv3x3_wd128"):
//   * workgroup = 4 waves = 128 output pixels x 128 output channels, every wave 128 pixels x 32 channels;
//   * per 32-channel chunk the input halo tile is staged global -> registers -> ds_write_b128 into four planes of [pixel][8 ch]
//     16-byte slots (double-buffered, one barrier per chunk); all nine taps read shifted windows of it;
//   * the wave's filter fragments go global -> VGPR from the packed fragment image (2 x 1 KiB per tap), two taps ahead;
//   * column-major tap order: the input-row fragments of one dx serve its three dy taps.
// M16: 8 rows x 16 pixels per wave, per dx 10 ds_read_b128 + 3 x 16 MFMA 16x16x32 (48 MFMAs, 768 pipe cycles)
// M32: 4 rows x 32 pixels per wave, per dx 12 ds_read_b128 + 3 x  8 MFMA 32x32x16 (24 MFMAs, 768 pipe cycles); the A fragment
//      (32 couts x 16 k) is read from the SAME packed image through a lane permutation (no repacking).
// The epilogue only reduces the accumulators to one store per lane (both variants alike): this measures the main loop.
// Output: us per launch, TFLOP/s for a few VGG16 / D shapes; run under rocprofv3 --pmc for MFMA busy / LDS conflicts and under
// tools/power_probe-style polling for watts.      build: hipcc --offload-arch=gfx950 -O3 -o mfma_shape mfma_shape.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#ifdef UB_BF16   // same loops on the bf16 instruction (operand bits are random either way): is the fp16 multiplier array the power hog?
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
#define __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, x, y, z) __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, x, y, z)
#define __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, x, y, z) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, x, y, z)
#endif

struct Args {
  const unsigned short* x;   // NHWC fp16, C channels, H x W, N images
  const unsigned short* w;   // packed [chunk][tap][tile16][lane][8]
  float* y;                  // [workgroups][256] sink
  int H, W, C, nchunk, ntile_total, tiles_x, tiles_y;
};

template <int M32>
struct Cfg {
  static constexpr int TH = M32 == 1 ? 4 : (M32 == 2 ? 16 : 8), TW = M32 == 1 ? 32 : 16;
  static constexpr int IH = TH + 2, IW = TW + 2, NPIX = IH * IW, NPIXR = (NPIX + 15) / 16 * 16;
  static constexpr int PLANE_B = NPIXR * 16, IN_BYTES = 4 * PLANE_B;
  static_assert(PLANE_B % 256 == 0, "plane stride must be a multiple of 256 B");
  static constexpr int UNITS = 4 * NPIXR, UPT = (UNITS + 255) / 256;
};

#ifndef UB_WAVES
#define UB_WAVES 2
#endif
// ABL (ablations of the M16 loop, results wrong): 1 filter fragments loaded once, 2 input fragments read from LDS once per chunk
// (dx = 0 only), 4 no staging of the next chunk and no barrier, 8 no s_setprio
template <int M32, int ABL = 0>
__global__ __launch_bounds__(256, UB_WAVES) void k(Args a) {
  using C = Cfg<M32>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int tile = blockIdx.x;
  const int tx = tile % a.tiles_x;
  tile /= a.tiles_x;
  const int ty = tile % a.tiles_y, n = tile / a.tiles_y;
  const int oy0 = ty * C::TH, ox0 = tx * C::TW;
  const unsigned short* xn = a.x + (long long)n * a.H * a.W * a.C;
  int goff[C::UPT];
  bool ok[C::UPT];
#pragma unroll
  for (int i = 0; i < C::UPT; ++i) {
    const int u = tid + i * 256, kg = u / C::NPIXR, p = u - kg * C::NPIXR;
    const int py = p / C::IW, px = p - py * C::IW;
    const int gy = oy0 - 1 + py, gx = ox0 - 1 + px;
    ok[i] = u < C::UNITS && p < C::NPIX && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
    goff[i] = ok[i] ? (gy * a.W + gx) * a.C + kg * 8 : 0;
  }
  u32x4 rin[C::UPT];
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  auto load_in = [&](int chunk) {
#pragma unroll
    for (int i = 0; i < C::UPT; ++i) rin[i] = *reinterpret_cast<const u32x4*>(xn + goff[i] + chunk * 32);
  };
  auto store_in = [&](char* buf) {
#pragma unroll
    for (int i = 0; i < C::UPT; ++i)
      if (tid + i * 256 < C::UNITS) *reinterpret_cast<u32x4*>(buf + (tid + i * 256) * 16) = ok[i] ? rin[i] : zero4;
  };
  const long long wstep = (long long)a.ntile_total * 512;   // elements per (chunk, tap)
  const int by = blockIdx.y;

  if constexpr (M32 == 2) {
    // PT16: 16 rows x 16 pixels x 32 couts per wave (128 accumulator registers): a filter fragment feeds 32 MFMAs instead of 16 --
    // half the L2 -> register filter bytes per MFMA -- the three taps of one dx are loaded once and serve both 8-row halves
    const int m = lane & 15, kgl = lane >> 4;
    f32x4 acc[16][2];
#pragma unroll
    for (int p = 0; p < 16; ++p) acc[p][0] = acc[p][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    const char* xfrag0 = smem + kgl * C::PLANE_B + m * 16;
    const unsigned short* wch = a.w + lane * 8 + (long long)(by * 8 + wave * 2) * 512;
    auto wload3 = [&](u32x4 (&dst)[3][2], const unsigned short* base, int dx) {
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int c = 0; c < 2; ++c) dst[dy][c] = *reinterpret_cast<const u32x4*>(base + (dy * 3 + dx) * wstep + c * 512);
    };
    load_in(0);
    store_in(smem);
    u32x4 wc[3][2];
    wload3(wc, wch, 0);
    __syncthreads();
    for (int chunk = 0; chunk < a.nchunk; ++chunk) {
      const bool has_next = chunk + 1 < a.nchunk;
      const unsigned short* wnext = wch + (has_next ? 9 * wstep : 0);
      const char* xb = xfrag0 + (chunk & 1) * C::IN_BYTES;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        // the next chunk's input tile in three pieces, each in registers for one dx phase only (its LDS buffer is free all chunk long)
        u32x4 rpc[2];
        const int nc = has_next ? chunk + 1 : chunk;
#pragma unroll
        for (int i = 0; i < 2; ++i) rpc[i] = *reinterpret_cast<const u32x4*>(xn + goff[2 * dx + i] + nc * 32);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          u32x4 xr[10];
#pragma unroll
          for (int r = 0; r < 10; ++r) xr[r] = *reinterpret_cast<const u32x4*>(xb + ((half * 8 + r) * C::IW + dx) * 16);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int dy = 0; dy < 3; ++dy) {
#pragma unroll
            for (int p = 0; p < 8; ++p)
#pragma unroll
              for (int c = 0; c < 2; ++c)
                acc[half * 8 + p][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wc[dy][c]), __builtin_bit_cast(f16x8, xr[p + dy]), acc[half * 8 + p][c], 0, 0, 0);
            if (half == 1) {   // this tap's fragments are dead: refill them with the same filter row of the next dx (two taps' MFMAs ahead of their use)
              const unsigned short* nb = dx + 1 < 3 ? wch : wnext;
              const int ndx = dx + 1 < 3 ? dx + 1 : 0;
              __builtin_amdgcn_sched_barrier(0);
#pragma unroll
              for (int c = 0; c < 2; ++c) wc[dy][c] = *reinterpret_cast<const u32x4*>(nb + (dy * 3 + ndx) * wstep + c * 512);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
          __builtin_amdgcn_sched_barrier(0);   // keep the other half's fragment reads below this half's MFMAs (40 registers, not 80)
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
          if (tid + (2 * dx + i) * 256 < C::UNITS)
            *reinterpret_cast<u32x4*>(smem + ((chunk + 1) & 1) * C::IN_BYTES + (tid + (2 * dx + i) * 256) * 16) = ok[2 * dx + i] ? rpc[i] : zero4;
      }
      wch = wnext;
      __syncthreads();
    }
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int p = 0; p < 16; ++p) s += acc[p][0] + acc[p][1] * 3.f;
    a.y[((long long)(blockIdx.y * gridDim.x + blockIdx.x)) * 256 + tid] = s[0] + 2.f * s[1] + 3.f * s[2] + 5.f * s[3];
  } else
  if constexpr (!M32) {
    const int m = lane & 15, kgl = lane >> 4;
    f32x4 acc[8][2];
#pragma unroll
    for (int p = 0; p < 8; ++p) acc[p][0] = acc[p][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    const char* xfrag0 = smem + kgl * C::PLANE_B + m * 16;
    const unsigned short* wch = a.w + lane * 8 + (long long)(by * 8 + wave * 2) * 512;
    auto wload = [&](u32x4 (&dst)[2], const unsigned short* base, int e) {
      const int tap = (e % 3) * 3 + (e / 3);
#pragma unroll
      for (int c = 0; c < 2; ++c) dst[c] = *reinterpret_cast<const u32x4*>(base + tap * wstep + c * 512);
    };
    load_in(0);
    store_in(smem);
    u32x4 wcur[2], wnx1[2], wnx2[2];
    wload(wcur, wch, 0);
    wload(wnx1, wch, 1);
    __syncthreads();
    u32x4 xr[10];
#pragma unroll
    for (int r = 0; r < 10; ++r) xr[r] = *reinterpret_cast<const u32x4*>(xfrag0 + (r * C::IW) * 16);
    for (int chunk = 0; chunk < a.nchunk; ++chunk) {
      const bool has_next = chunk + 1 < a.nchunk;
      if (has_next && !(ABL & 4)) load_in(chunk + 1);
      const unsigned short* wnext = wch + (has_next ? 9 * wstep : 0);
      const char* xb = xfrag0 + (chunk & 1) * C::IN_BYTES;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        if (!(ABL & 2) || dx == 0) {
#pragma unroll
          for (int r = 0; r < 10; ++r) xr[r] = *reinterpret_cast<const u32x4*>(xb + (r * C::IW + dx) * 16);
        }
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          const int e = dx * 3 + dy;
          if (!(ABL & 1)) {
            if (e + 2 < 9) wload(wnx2, wch, e + 2);
            else wload(wnx2, wnext, e + 2 - 9);
          }
          __builtin_amdgcn_sched_barrier(0);
          if (!(ABL & 8)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
          for (int p = 0; p < 8; ++p)
#pragma unroll
            for (int c = 0; c < 2; ++c)
              acc[p][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wcur[c]), __builtin_bit_cast(f16x8, xr[p + dy]), acc[p][c], 0, 0, 0);
          if (!(ABL & 8)) __builtin_amdgcn_s_setprio(0);
          if (!(ABL & 1)) {
#pragma unroll
            for (int c = 0; c < 2; ++c) wcur[c] = wnx1[c], wnx1[c] = wnx2[c];
          } else {
            asm volatile("" : "+v"(wcur[0]), "+v"(wcur[1]));
          }
        }
      }
      wch = wnext;
      if (!(ABL & 4)) {
        if (has_next) store_in(smem + ((chunk + 1) & 1) * C::IN_BYTES);
        __syncthreads();
      }
    }
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int p = 0; p < 8; ++p) s += acc[p][0] + acc[p][1] * 3.f;
    a.y[((long long)(blockIdx.y * gridDim.x + blockIdx.x)) * 256 + tid] = s[0] + 2.f * s[1] + 3.f * s[2] + 5.f * s[3];
  } else {
    const int m = lane & 31, kq = lane >> 5;     // pixel column, 8-channel group inside a 16-channel half
    f32x16 acc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[r][q] = 0.f;
    // B fragment of half h: plane 2 h + kq
    const char* xfrag0 = smem + kq * C::PLANE_B + m * 16;
    // A fragment (32 couts x 16 k, half h) out of the packed 2 x (16 cout x 32 k) image: cout = lane & 31, k = 16 h + 8 kq + j
    const unsigned short* wch = a.w + (long long)(by * 8 + wave * 2 + ((lane & 31) >> 4)) * 512 + (kq * 16 + (lane & 15)) * 8;
    auto wload = [&](u32x4 (&dst)[2], const unsigned short* base, int e) {
      const int tap = (e % 3) * 3 + (e / 3);
#pragma unroll
      for (int h = 0; h < 2; ++h) dst[h] = *reinterpret_cast<const u32x4*>(base + tap * wstep + h * 256);   // half h: lanes' k-groups 2 h, 2 h + 1
    };
    load_in(0);
    store_in(smem);
    u32x4 wcur[2], wnx1[2], wnx2[2];
    wload(wcur, wch, 0);
    wload(wnx1, wch, 1);
    __syncthreads();
    for (int chunk = 0; chunk < a.nchunk; ++chunk) {
      const bool has_next = chunk + 1 < a.nchunk;
      if (has_next) load_in(chunk + 1);
      const unsigned short* wnext = wch + (has_next ? 9 * wstep : 0);
      const char* xb = xfrag0 + (chunk & 1) * C::IN_BYTES;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        u32x4 xr[2][6];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int r = 0; r < 6; ++r) xr[h][r] = *reinterpret_cast<const u32x4*>(xb + h * 2 * C::PLANE_B + (r * C::IW + dx) * 16);
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          const int e = dx * 3 + dy;
          if (e + 2 < 9) wload(wnx2, wch, e + 2);
          else wload(wnx2, wnext, e + 2 - 9);
          __builtin_amdgcn_sched_barrier(0);
          __builtin_amdgcn_s_setprio(1);
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wcur[h]), __builtin_bit_cast(f16x8, xr[h][r + dy]), acc[r], 0, 0, 0);
          __builtin_amdgcn_s_setprio(0);
#pragma unroll
          for (int h = 0; h < 2; ++h) wcur[h] = wnx1[h], wnx1[h] = wnx2[h];
        }
      }
      wch = wnext;
      if (has_next) store_in(smem + ((chunk + 1) & 1) * C::IN_BYTES);
      __syncthreads();
    }
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int q = 0; q < 16; ++q) s += acc[r][q] * (float)(1 + (q & 3));
    a.y[((long long)(blockIdx.y * gridDim.x + blockIdx.x)) * 256 + tid] = s;
  }
}

// CPU check of one workgroup's sink value is not attempted: both variants compute the same convolution, so the SUM of all outputs
// weighted identically would differ by layout; instead the two are validated against each other on a per-(pixel, cout) basis
// by a third, direct kernel below on a tiny shape.
__global__ void direct(const unsigned short* x, const unsigned short* wdense, float* out, int N, int H, int W, int C, int CO) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  if (i >= (long long)N * H * W * CO) return;
  const int co = i % CO;
  long long r = i / CO;
  const int px = r % W;
  r /= W;
  const int py = r % H, n = r / H;
  float s = 0.f;
  for (int dy = 0; dy < 3; ++dy)
    for (int dx = 0; dx < 3; ++dx) {
      const int gy = py - 1 + dy, gx = px - 1 + dx;
      if (gy < 0 || gy >= H || gx < 0 || gx >= W) continue;
      for (int c = 0; c < C; ++c)
        s += (float)reinterpret_cast<const _Float16*>(x)[((long long)(n * H + gy) * W + gx) * C + c] *
             (float)reinterpret_cast<const _Float16*>(wdense)[((co * 9 + dy * 3 + dx) * (long long)C) + c];
    }
  out[i] = s;
}

// full-output variants of the two kernels for the correctness check (store acc per pixel / cout) -- same loops, real epilogue
template <int M32>
__global__ __launch_bounds__(256, 2) void kfull(Args a, float* out, int CO) {
  // re-run k<M32>'s arithmetic with a plain store of every accumulator element: implemented by recomputing through the same code path
  // is not possible without duplicating it, so the check kernel is a straightforward re-statement of the fragment maps:
  using C = Cfg<M32>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int tile = blockIdx.x;
  const int tx = tile % a.tiles_x;
  tile /= a.tiles_x;
  const int ty = tile % a.tiles_y, n = tile / a.tiles_y;
  const int oy0 = ty * C::TH, ox0 = tx * C::TW;
  const unsigned short* xn = a.x + (long long)n * a.H * a.W * a.C;
  const long long wstep = (long long)a.ntile_total * 512;
  const int by = blockIdx.y;
  f32x16 acc32[4];
  f32x4 acc16[8][2];
  for (int r = 0; r < 4; ++r) for (int q = 0; q < 16; ++q) acc32[r][q] = 0.f;
  for (int p = 0; p < 8; ++p) acc16[p][0] = acc16[p][1] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int chunk = 0; chunk < a.nchunk; ++chunk) {
    __syncthreads();
    for (int u = tid; u < C::UNITS; u += 256) {
      const int kg = u / C::NPIXR, p = u - kg * C::NPIXR, py = p / C::IW, px = p - py * C::IW;
      const int gy = oy0 - 1 + py, gx = ox0 - 1 + px;
      const bool okk = p < C::NPIX && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (okk) v = *reinterpret_cast<const u32x4*>(xn + ((long long)gy * a.W + gx) * a.C + chunk * 32 + kg * 8);
      *reinterpret_cast<u32x4*>(smem + u * 16) = v;
    }
    __syncthreads();
    for (int dy = 0; dy < 3; ++dy)
      for (int dx = 0; dx < 3; ++dx) {
        const int tap = dy * 3 + dx;
        const unsigned short* wb = a.w + ((long long)chunk * 9 + tap) * wstep;
        if constexpr (M32) {
          const int m = lane & 31, kq = lane >> 5;
          for (int h = 0; h < 2; ++h) {
            const u32x4 wf = *reinterpret_cast<const u32x4*>(wb + (long long)(by * 8 + wave * 2 + ((lane & 31) >> 4)) * 512 + (kq * 16 + (lane & 15)) * 8 + h * 256);
            for (int r = 0; r < 4; ++r) {
              const u32x4 xf = *reinterpret_cast<const u32x4*>(smem + (2 * h + kq) * C::PLANE_B + ((r + dy) * C::IW + m + dx) * 16);
              acc32[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wf), __builtin_bit_cast(f16x8, xf), acc32[r], 0, 0, 0);
            }
          }
        } else {
          const int m = lane & 15, kgl = lane >> 4;
          for (int c = 0; c < 2; ++c) {
            const u32x4 wf = *reinterpret_cast<const u32x4*>(wb + (long long)(by * 8 + wave * 2 + c) * 512 + lane * 8);
            for (int p = 0; p < 8; ++p) {
              const u32x4 xf = *reinterpret_cast<const u32x4*>(smem + kgl * C::PLANE_B + ((p + dy) * C::IW + m + dx) * 16);
              acc16[p][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wf), __builtin_bit_cast(f16x8, xf), acc16[p][c], 0, 0, 0);
            }
          }
        }
      }
  }
  if constexpr (M32) {
    const int m = lane & 31;
    for (int r = 0; r < 4; ++r)
      for (int q = 0; q < 16; ++q) {
        const int co = by * 128 + wave * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
        const int oy = oy0 + r, ox = ox0 + m;
        if (oy < a.H && ox < a.W && co < CO) out[(((long long)n * a.H + oy) * a.W + ox) * CO + co] = acc32[r][q];
      }
  } else {
    const int m = lane & 15, kgl = lane >> 4;
    for (int p = 0; p < 8; ++p)
      for (int c = 0; c < 2; ++c)
        for (int q = 0; q < 4; ++q) {
          const int co = by * 128 + wave * 32 + c * 16 + kgl * 4 + q;
          const int oy = oy0 + p, ox = ox0 + m;
          if (oy < a.H && ox < a.W && co < CO) out[(((long long)n * a.H + oy) * a.W + ox) * CO + co] = acc16[p][c][q];
        }
  }
}

static unsigned short f2h(float f) { _Float16 h = (_Float16)f; unsigned short u; memcpy(&u, &h, 2); return u; }

int main(int argc, char** argv) {
  // ---- correctness of both fragment maps on a small shape against the direct kernel
  {
    const int N = 1, H = 16, W = 32, Cc = 64, CO = 128, nch = 2, ntile = 8;
    std::vector<unsigned short> hx((size_t)N * H * W * Cc), hwd((size_t)CO * 9 * Cc), hwp((size_t)nch * 9 * ntile * 512);
    srand(1);
    for (auto& v : hx) v = f2h((rand() % 2001 - 1000) / 1000.f);
    for (auto& v : hwd) v = f2h((rand() % 2001 - 1000) / 4000.f);
    for (int ch = 0; ch < nch; ++ch)
      for (int tap = 0; tap < 9; ++tap)
        for (int t = 0; t < ntile; ++t)
          for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 8; ++j) {
              const int co = t * 16 + (l & 15), k = (l >> 4) * 8 + j, c = ch * 32 + k;
              hwp[(((size_t)ch * 9 + tap) * ntile + t) * 512 + l * 8 + j] = hwd[((size_t)co * 9 + tap) * Cc + c];
            }
    unsigned short *dx, *dwd, *dwp;
    float *o0, *o1, *o2;
    const size_t no = (size_t)N * H * W * CO;
    CK(hipMalloc(&dx, hx.size() * 2)); CK(hipMalloc(&dwd, hwd.size() * 2)); CK(hipMalloc(&dwp, hwp.size() * 2));
    CK(hipMalloc(&o0, no * 4)); CK(hipMalloc(&o1, no * 4)); CK(hipMalloc(&o2, no * 4));
    CK(hipMemcpy(dx, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dwd, hwd.data(), hwd.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dwp, hwp.data(), hwp.size() * 2, hipMemcpyHostToDevice));
    direct<<<(unsigned)((no + 255) / 256), 256>>>(dx, dwd, o0, N, H, W, Cc, CO);
    Args a16{dx, dwp, nullptr, H, W, Cc, nch, ntile, W / 16, H / 8}, a32{dx, dwp, nullptr, H, W, Cc, nch, ntile, W / 32, H / 4};
    kfull<0><<<dim3(a16.tiles_x * a16.tiles_y * N, 1), 256, 2 * Cfg<0>::IN_BYTES>>>(a16, o1, CO);
    kfull<1><<<dim3(a32.tiles_x * a32.tiles_y * N, 1), 256, 2 * Cfg<1>::IN_BYTES>>>(a32, o2, CO);
    std::vector<float> h0(no), h1(no), h2(no);
    CK(hipMemcpy(h0.data(), o0, no * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h1.data(), o1, no * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h2.data(), o2, no * 4, hipMemcpyDeviceToHost));
    double e1 = 0, e2 = 0, sc = 0;
    for (size_t i = 0; i < no; ++i) {
      e1 = fmax(e1, fabs(h1[i] - h0[i])); e2 = fmax(e2, fabs(h2[i] - h0[i])); sc = fmax(sc, fabs(h0[i]));
    }
    printf("# fragment-map check vs direct conv: max |16x16x32 - ref| %.3g, max |32x32x16 - ref| %.3g (scale %.3g)\n", e1, e2, sc);
  }
  // ---- timing
  struct Shape { const char* name; int N, H, W, C, CO; };
  const Shape shapes[] = {{"VGG16 conv2_2 128->128 @128^2", 16, 128, 128, 128, 128}, {"VGG16 conv3_2 256->256 @64^2", 16, 64, 64, 256, 256},
                          {"VGG16 conv4_2 512->512 @32^2", 16, 32, 32, 512, 512}, {"refine 160->128 @128^2", 16, 128, 128, 160, 128}};
  const int reps = argc > 1 ? atoi(argv[1]) : 50;
  const int only = argc > 2 ? atoi(argv[2]) : -1;   // -1 both; 0 / 1: one variant (for PMC / power runs)
  for (const Shape& s : shapes) {
    const int nch = s.C / 32, ntile = s.CO / 16;
    const size_t nx = (size_t)s.N * s.H * s.W * s.C, nw = (size_t)nch * 9 * ntile * 512;
    std::vector<unsigned short> hx(nx), hw(nw);
    for (auto& v : hx) v = f2h((rand() % 2001 - 1000) / 1000.f);
    for (auto& v : hw) v = f2h((rand() % 2001 - 1000) / 4000.f);
    unsigned short *dx, *dw;
    float* dy;
    CK(hipMalloc(&dx, nx * 2)); CK(hipMalloc(&dw, nw * 2));
    CK(hipMemcpy(dx, hx.data(), nx * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, hw.data(), nw * 2, hipMemcpyHostToDevice));
    const double flop = 2.0 * s.N * s.H * s.W * (double)s.C * s.CO * 9;
    static const char* vname[] = {"16x16x32", "32x32x16", "16: filter once", "16: x frags 1/3", "16: no staging", "16: no setprio", "16: MFMA only", "16: no operand ld", "16: 16 rows/wave"};
    for (int v = 0; v < 9; ++v) {
      if (only >= 0 && only != v) continue;
      Args a{dx, dw, nullptr, s.H, s.W, s.C, nch, ntile, v == 1 ? s.W / 32 : s.W / 16, v == 1 ? s.H / 4 : (v == 8 ? s.H / 16 : s.H / 8)};
      const dim3 grid(a.tiles_x * a.tiles_y * s.N, s.CO / 128);
      CK(hipMalloc(&dy, (size_t)grid.x * grid.y * 256 * 4));
      a.y = dy;
      const unsigned lds = 2 * (v == 1 ? Cfg<1>::IN_BYTES : (v == 8 ? Cfg<2>::IN_BYTES : Cfg<0>::IN_BYTES));
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      float best = 1e30f, tot = 0.f;
      for (int round = 0; round < 3; ++round) {
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) {
          if (v == 1) k<1><<<grid, 256, lds>>>(a);
          else if (v == 0) k<0><<<grid, 256, lds>>>(a);
          else if (v == 2) k<0, 1><<<grid, 256, lds>>>(a);
          else if (v == 3) k<0, 2><<<grid, 256, lds>>>(a);
          else if (v == 4) k<0, 4><<<grid, 256, lds>>>(a);
          else if (v == 5) k<0, 8><<<grid, 256, lds>>>(a);
          else if (v == 6) k<0, 7><<<grid, 256, lds>>>(a);
          else if (v == 7) k<0, 3><<<grid, 256, lds>>>(a);
          else k<2><<<grid, 256, lds>>>(a);
        }
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = fminf(best, ms / reps);
        tot = ms / reps;
      }
      printf("%-34s %-18s grid %5u x %u  %8.2f us (best of 3; last %8.2f)  %7.1f TFLOP/s\n", s.name, vname[v], grid.x, grid.y, best * 1e3, tot * 1e3,
             flop / (best * 1e-3) * 1e-12);
      CK(hipFree(dy));
    }
    CK(hipFree(dx)); CK(hipFree(dw));
  }
  return 0;
}
