// raggedrow.hip -- what a channel PREFIX of a pixel-major buffer costs in HBM time (experiment aid; VERDICT r5 #1a).
//   hipcc --offload-arch=gfx950 -O3 -w -o /tmp/raggedrow tools/ubench/raggedrow.hip && /tmp/raggedrow
// The dense layers' kernels read x[p][0 .. C) and read-modify-write G[p][0 .. C) of [pixels][pitch] buffers with C = 64 + 32 i.
// The fused bottleneck backward takes as long for C = 96 as for C = 128 (and 160 as 192, 224 as 256): "a channel tile costs its
// step count".  This streaming kernel has NO tiles, steps, LDS or MFMA -- G[p][c] += a * x[p][c] on 16-byte pieces, eight pieces
// in flight per lane -- and shows the same staircase: the memory system moves 128-byte lines, a row that ends in half a line
// costs the whole line, and the time of a prefix is the time of the prefix rounded up to 64 channels.  The second table is the
// same work in a channel-blocked layout [C / 32][pixels][32] (every 32-channel group contiguous over pixels): no staircase.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef unsigned short u16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int NP, bool BLOCKED>   // NP = C / 8 pieces per pixel
__global__ __launch_bounds__(256) void axpy_prefix(const u16* __restrict__ x, u16* __restrict__ g, long long P, int pitch, float a) {
  const long long total = P * NP, stride = (long long)gridDim.x * 256;
  auto addr = [&](long long q) -> long long {
    const long long p = q / NP;
    const int j = (int)(q - p * NP);
    return BLOCKED ? ((long long)(j >> 2) * P + p) * 32 + (j & 3) * 8 : p * pitch + j * 8;
  };
  for (long long q0 = (long long)blockIdx.x * 256 + threadIdx.x; q0 < total; q0 += 4 * stride) {
    u32x4 xv[4], gv[4];
    long long o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const long long q = q0 + k * stride < total ? q0 + k * stride : q0;
      o[k] = addr(q);
      xv[k] = *reinterpret_cast<const u32x4*>(x + o[k]);
      gv[k] = *reinterpret_cast<const u32x4*>(g + o[k]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      u32x4 r;
#pragma unroll
      for (int e = 0; e < 4; ++e) {   // two bf16 per dword
        const float g0 = __uint_as_float(gv[k][e] << 16), g1 = __uint_as_float(gv[k][e] & 0xffff0000u);
        const float x0 = __uint_as_float(xv[k][e] << 16), x1 = __uint_as_float(xv[k][e] & 0xffff0000u);
        r[e] = (__float_as_uint(g0 + a * x0) >> 16) | (__float_as_uint(g1 + a * x1) & 0xffff0000u);
      }
      if (q0 + k * stride < total) *reinterpret_cast<u32x4*>(g + o[k]) = r;
    }
  }
}

template <int NP, bool BLOCKED>
float run(const u16* x, u16* g, long long P, int pitch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  float best = 1e9f;
  for (int it = 0; it < 6; ++it) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((axpy_prefix<NP, BLOCKED>), dim3(256 * 8), dim3(256), 0, 0, x, g, P, pitch, 0.001f);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (it > 0 && ms < best) best = ms;
  }
  return best * 1e3f;
}

int main() {
  const long long P = 16ll * 256 * 256;
  const int pitch = 256;
  u16 *x, *g;
  hipMalloc(&x, P * pitch * 2), hipMalloc(&g, P * pitch * 2);
  hipMemset(x, 0x11, P * pitch * 2), hipMemset(g, 0x22, P * pitch * 2);
  printf("G[p][0..C) += a x[p][0..C), bf16, P = 16 x 256 x 256 pixels; bytes = 3 P C 2 (two reads + one write)\n");
  for (int blocked = 0; blocked < 2; ++blocked) {
    printf(blocked ? "\nchannel-blocked layout [C/32][pixels][32]\n" : "\npixel-major layout, pitch %d channels (%d B)\n", pitch, pitch * 2);
    printf("%6s %10s %18s %26s\n", "C", "us", "TB/s algorithmic", "TB/s of 128-byte lines");
    const int cs[7] = {64, 96, 128, 160, 192, 224, 256};
    for (int i = 0; i < 7; ++i) {
      const int c = cs[i];
      float us = 0.f;
#define CASE(N)                                                                                   \
  case N: us = blocked ? run<N / 8, true>(x, g, P, pitch) : run<N / 8, false>(x, g, P, pitch); \
    break;
      switch (c) { CASE(64) CASE(96) CASE(128) CASE(160) CASE(192) CASE(224) CASE(256) }
      const double alg = 3.0 * P * c * 2, lines = 3.0 * P * ((c + 63) / 64 * 64) * 2;
      printf("%6d %10.1f %18.2f %26.2f\n", c, us, alg / us / 1e6, (blocked ? alg : lines) / us / 1e6);
    }
  }
  return 0;
}
