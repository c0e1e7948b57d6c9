// Access-pattern micro-benchmark (tuning aid, not product code): what HBM rate do the
// load / store shapes of the conv kernels reach on their own?
//   hipcc --offload-arch=gfx950 -O3 -o patterns patterns.hip && ./patterns
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// P pixels, pitch_in / pitch_out in bytes per pixel; read cin_b bytes, write cout_b bytes per pixel.
// mode bit0: loads fragment-shaped (lane m=l&15 pixel, g=l>>4 -> 32 B) vs linear (lane*16 over pixel rows)
// mode bit1: stores 8 B/lane fragment-shaped (16 px x 4 groups) vs linear 16 B/lane
template <int LOADFRAG, int STOREFRAG>
__global__ __launch_bounds__(256) void k(const char* x, char* y, long long P, int pin, int pout, int cin_b, int cout_b, int ntiles) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m = lane & 15, g = lane >> 4;
  u32x4 accv = {0, 0, 0, 0};
  for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long long px0 = tile * 256 + wave * 64;
    if (LOADFRAG) {
      for (int kb = 0; kb < cin_b; kb += 128)
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(x + (px0 + p * 16 + m) * pin + kb + g * 32 + j * 16);
            accv += v;
          }
    } else {   // 64 px * cin_b bytes, linear: each 8-lane group reads one pixel's 128 B
      for (int kb = 0; kb < cin_b; kb += 128)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const u32x4 v = *reinterpret_cast<const u32x4*>(x + (px0 + i * 8 + (lane >> 3)) * pin + kb + (lane & 7) * 16);
          accv += v;
        }
    }
    // make the stored data depend on the loads
    const unsigned s = accv[0] ^ accv[1] ^ accv[2] ^ accv[3];
    if (STOREFRAG) {
      for (int cb = 0; cb < cout_b; cb += 32)   // one 16-cout tile = 32 B per pixel; lane writes 8 B
#pragma unroll
        for (int p = 0; p < 4; ++p)
          *reinterpret_cast<u32x2*>(y + (px0 + p * 16 + m) * pout + cb + g * 8) = u32x2{s, s + 1};
    } else {
      for (int cb = 0; cb < cout_b; cb += 128)
#pragma unroll
        for (int i = 0; i < 8; ++i)
          *reinterpret_cast<u32x4*>(y + (px0 + i * 8 + (lane >> 3)) * pout + cb + (lane & 7) * 16) = u32x4{s, s, s, s};
    }
  }
}

template <int LF, int SF>
void run(const char* name, const char* x, char* y, long long P, int pin, int pout, int cin_b, int cout_b, int grid) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int ntiles = (int)(P / 256);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<LF, SF>), dim3(grid), dim3(256), 0, 0, x, y, P, pin, pout, cin_b, cout_b, ntiles);
  CK(hipEventRecord(e0, 0));
  const int reps = 10;
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k<LF, SF>), dim3(grid), dim3(256), 0, 0, x, y, P, pin, pout, cin_b, cout_b, ntiles);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / reps, gb = (double)P * (cin_b + cout_b) / 1e9;
  printf("%-34s grid %5d  read %4d B/px write %4d B/px : %8.1f us  %7.1f GB/s\n", name, grid, cin_b, cout_b, us, gb / (us * 1e-6));
}

int main() {
  const long long P = 16ll * 256 * 256;
  char *x, *y;
  CK(hipMalloc(&x, P * 512)); CK(hipMalloc(&y, P * 512));
  CK(hipMemset(x, 1, P * 512)); CK(hipMemset(y, 0, P * 512));
  for (int grid : {512, 1024, 2048, 4096}) {
    run<1, 1>("frag loads + frag 8B stores", x, y, P, 512, 256, 256, 256, grid);
    run<0, 1>("linear loads + frag 8B stores", x, y, P, 512, 256, 256, 256, grid);
    run<1, 0>("frag loads + linear 16B stores", x, y, P, 512, 256, 256, 256, grid);
    run<0, 0>("linear loads + linear stores", x, y, P, 512, 256, 256, 256, grid);
  }
  run<1, 1>("frag/frag read-only-ish", x, y, P, 512, 256, 256, 0, 1024);
  run<0, 0>("linear read only", x, y, P, 512, 256, 256, 0, 1024);
  run<1, 1>("frag write only (8B)", x, y, P, 512, 256, 0, 256, 1024);
  run<0, 0>("linear write only (16B)", x, y, P, 512, 256, 0, 256, 1024);
  run<1, 1>("frag write 64B slice of 512B pitch", x, y, P, 512, 512, 256, 64, 1024);
  run<0, 0>("dense pitch r256/w256 linear", x, y, P, 256, 256, 256, 256, 1024);
  return 0;
}
