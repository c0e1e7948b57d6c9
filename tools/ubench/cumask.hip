// cumask.hip -- helper of tools/cu_mask_sweep.py (experiment aid, not product): streams with a CU mask, a census of where their
// workgroups really run, and a plain streaming copy as the HBM reference curve vs CU count.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o /tmp/libcumask.so tools/ubench/cumask.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

extern "C" void* cumask_stream_create(int nwords, const uint32_t* mask) {
  hipStream_t s = nullptr;
  if (hipExtStreamCreateWithCUMask(&s, (uint32_t)nwords, mask) != hipSuccess) return nullptr;
  return s;
}
extern "C" int cumask_stream_get(void* s, int nwords, uint32_t* mask) { return (int)hipExtStreamGetCUMask((hipStream_t)s, (uint32_t)nwords, mask); }
extern "C" void cumask_stream_destroy(void* s) { (void)hipStreamDestroy((hipStream_t)s); }

// one record per workgroup: HW_REG_HW_ID (cu_id [11:8], sh_id [12], se_id [15:13] on gfx9) and HW_REG_XCC_ID
__global__ void census_kernel(uint32_t* out, int spin) {
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    out[2 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
  }
  for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(64);   // stay resident so that later workgroups spread over the other CUs
}
extern "C" int cumask_census(void* s, uint32_t* out, int nwg, int spin) {
  // 64 KB of LDS per workgroup: at most two per CU, so 2 x (CUs of the mask) are resident at once
  hipLaunchKernelGGL(census_kernel, dim3(nwg), dim3(64), 64 * 1024, (hipStream_t)s, out, spin);
  return (int)hipGetLastError();
}

// streaming copy, 16 B per lane, grid-stride: `nwg` workgroups of 256 threads
__global__ __launch_bounds__(256) void copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, long long n16) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long stride = (long long)gridDim.x * 256;
  for (; i + 3 * stride < n16; i += 4 * stride) {
    const uint4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
    dst[i] = a, dst[i + stride] = b, dst[i + 2 * stride] = c, dst[i + 3 * stride] = d;
  }
  for (; i < n16; i += stride) dst[i] = src[i];
}
extern "C" int cumask_copy(void* s, const void* src, void* dst, long long bytes, int nwg) {
  hipLaunchKernelGGL(copy_kernel, dim3(nwg), dim3(256), 0, (hipStream_t)s, (const uint4*)src, (uint4*)dst, bytes / 16);
  return (int)hipGetLastError();
}
