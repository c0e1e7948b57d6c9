// K-streamed reads of a pixel tile through LDS-DMA: each workgroup owns PX consecutive pixels of a
// [npix][pitch B] tensor and walks along the channel axis in pieces of PIECE bytes per pixel (the read
// pattern of a 1x1 convolution that keeps its accumulators and streams K).  How does the achieved
// bandwidth depend on the piece size when the pixel pitch is 2 KiB (dense_block3's concat buffer)?
// Tuning aid for the x-stream 1x1 kernel.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ __forceinline__ void dma16(const char* g, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// PX pixels per workgroup, 8 waves; a stage = PX x PIECE bytes = PX*PIECE/1024 wave-instructions, spread over the waves.
template <int PX, int PIECE, int DEPTH>
__global__ __launch_bounds__(512) void k(const char* x, unsigned* sink, int pitch, int used) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr int STAGE_B = PX * PIECE, IPW = STAGE_B / 1024 / 8;   // instructions per wave per stage
  static_assert(IPW >= 1, "stage too small");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const char* base = x + (size_t)blockIdx.x * PX * pitch;
  const int nstage = used / PIECE;
  constexpr int LPP = PIECE / 16;   // lanes per pixel
  u32x4 acc = {0, 0, 0, 0};
  auto issue = [&](int s) {
    char* dst = lds + (s % (DEPTH + 1)) * STAGE_B;
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
      const int u = (wave * IPW + i) * 64 + lane, p = u / LPP, c = u % LPP;
      dma16(base + (size_t)p * pitch + (size_t)s * PIECE + c * 16, dst + (wave * IPW + i) * 1024);
    }
  };
  for (int s = 0; s < DEPTH && s < nstage; ++s) issue(s);
  for (int s = 0; s < nstage; ++s) {
    if (s + DEPTH < nstage) {
      issue(s + DEPTH);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IPW * DEPTH) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const char* src = lds + (s % (DEPTH + 1)) * STAGE_B;
#pragma unroll
    for (int i = 0; i < IPW; ++i) acc += *reinterpret_cast<const u32x4*>(src + (wave * IPW + i) * 1024 + lane * 16);
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345) sink[0] = 1;
}

template <int PX, int PIECE, int DEPTH>
void run(const char* x, unsigned* sink, size_t npix, int pitch, int used, hipEvent_t e0, hipEvent_t e1) {
  const int grid = (int)(npix / PX);
  const size_t lds = (size_t)(DEPTH + 1) * PX * PIECE;
  if (lds > 160 * 1024) return;
  CK(hipFuncSetAttribute((const void*)k<PX, PIECE, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  for (int w = 0; w < 2; ++w) {
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<PX, PIECE, DEPTH>), dim3(grid), dim3(512), lds, 0, x, sink, pitch, used);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double mb = (double)npix * (used / PIECE * PIECE) / 1e6;
    if (w)
      printf("npix %7zu pitch %4d used %4d B/px  tile %3d px  piece %4d B  depth %d (%5.1f KiB in flight/CU)  grid %4d: %7.1f us  %6.0f GB/s\n",
             npix, pitch, used, PX, PIECE, DEPTH, DEPTH * PX * PIECE / 1024.0, grid, ms * 200, mb / (ms * 0.2e-3) / 1e3);
  }
}

int main() {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  unsigned* sink;
  CK(hipMalloc(&sink, 4));
  {  // dense_block3: 16 x 64 x 64 pixels, 1024-channel concat buffer, read the first 640 / 1024 channels
    const size_t npix = 65536;
    const int pitch = 2048;
    char* x;
    CK(hipMalloc(&x, npix * pitch));
    CK(hipMemset(x, 1, npix * pitch));
    for (int used : {1280, 2048}) {
      run<256, 128, 2>(x, sink, npix, pitch, used, e0, e1);
      run<256, 128, 3>(x, sink, npix, pitch, used, e0, e1);
      run<256, 256, 1>(x, sink, npix, pitch, used, e0, e1);
      run<256, 256, 2>(x, sink, npix, pitch, used, e0, e1);
      run<128, 256, 2>(x, sink, npix, pitch, used, e0, e1);
      run<128, 256, 4>(x, sink, npix, pitch, used, e0, e1);
      run<128, 512, 2>(x, sink, npix, pitch, used, e0, e1);
      run<64, 512, 2>(x, sink, npix, pitch, used, e0, e1);
      run<64, 512, 4>(x, sink, npix, pitch, used, e0, e1);
      run<64, 1024, 2>(x, sink, npix, pitch, used < 2048 ? 1024 : 2048, e0, e1);
    }
    CK(hipFree(x));
  }
  {  // dense_block1: 16 x 256 x 256 pixels, 256-channel buffer (512 B pitch), read 128 / 192 channels
    const size_t npix = 1048576;
    const int pitch = 512;
    char* x;
    CK(hipMalloc(&x, npix * pitch));
    CK(hipMemset(x, 1, npix * pitch));
    for (int used : {256, 384}) {
      run<256, 128, 2>(x, sink, npix, pitch, used, e0, e1);
      run<256, 128, 3>(x, sink, npix, pitch, used, e0, e1);
      run<128, 128, 4>(x, sink, npix, pitch, used, e0, e1);
      if (used % 256 == 0) run<128, 256, 2>(x, sink, npix, pitch, used, e0, e1);
      if (used % 256 == 0) run<256, 256, 1>(x, sink, npix, pitch, used, e0, e1);
    }
    CK(hipFree(x));
  }
  return 0;
}
