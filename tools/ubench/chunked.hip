// Does reading a 256-byte-per-pixel tensor in 4 passes of 64 B per pixel (the 32-channel chunks of the
// LDS-tiled 3x3 conv) cost HBM efficiency?  (tuning aid)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
// each wave: 16 pixels x 64 B per instruction (lane = pixel*4 + g); bytes_per_px read per pixel starting at off
__global__ __launch_bounds__(256) void k(const char* x, unsigned* sink, long long P, int pitch, int off, int bytes_per_px, int inner_chunks) {
  const int lane = threadIdx.x & 63;
  u32x4 acc = {0, 0, 0, 0};
  const long long nblk = P / 16;   // groups of 16 pixels
  for (long long gidx = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); gidx < nblk; gidx += (long long)gridDim.x * 4) {
    const char* p = x + (gidx * 16 + (lane >> 2)) * pitch + off + (lane & 3) * 16;
    for (int c = 0; c < inner_chunks; ++c) acc += *reinterpret_cast<const u32x4*>(p + c * 64);
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345) sink[0] = 1;
}
int main() {
  const long long P = 16ll * 256 * 256;
  char* x; unsigned* sink;
  CK(hipMalloc(&x, P * 256)); CK(hipMalloc(&sink, 4)); CK(hipMemset(x, 1, P * 256));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int mode = 0; mode < 3; ++mode) {
    for (int w = 0; w < 2; ++w) {
      CK(hipEventRecord(e0, 0));
      const int reps = 5;
      for (int r = 0; r < reps; ++r) {
        if (mode == 0) hipLaunchKernelGGL(k, dim3(2048), dim3(256), 0, 0, x, sink, P, 256, 0, 256, 4);       // all 256 B per pixel in one pass
        else if (mode == 1) for (int c = 0; c < 4; ++c) hipLaunchKernelGGL(k, dim3(2048), dim3(256), 0, 0, x, sink, P, 256, c * 64, 64, 1);  // 4 passes x 64 B
        else hipLaunchKernelGGL(k, dim3(2048), dim3(256), 0, 0, x, sink, P, 64, 0, 64, 1);                       // 64 B per pixel, dense (planar chunk)
      }
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const double bytes = mode == 2 ? (double)P * 64 : (double)P * 256;
      if (w) printf("%s: %.1f us per full read, %.1f GB/s\n", mode == 0 ? "256 B/px one pass" : mode == 1 ? "4 passes x 64 B/px (pitch 256)" : "64 B/px dense plane",
                    ms * 1e3 / reps, bytes / (ms * 1e-3 / reps) / 1e9);
    }
  }
  return 0;
}
