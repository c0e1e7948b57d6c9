#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned short* in, unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = in[i];
  __syncthreads();
  // lane l supplies address: group g = l >> 4, i = l & 15: row r = i >> 2 (k), piece q = i & 3 -> &lds[g*1024 + r*64 + q*4]  (row pitch 64 elements)
  const int l = threadIdx.x, g = l >> 4, i = l & 15;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + g * 1024 + (i >> 2) * 64 + (i & 3) * 4));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)v[j];
}
int main() {
  unsigned short h[4096], o[256];
  for (int i = 0; i < 4096; ++i) h[i] = i;
  unsigned short *d, *od;
  hipMalloc(&d, sizeof(h)); hipMalloc(&od, sizeof(o));
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  k<<<1, 64>>>(d, od);
  hipMemcpy(o, od, sizeof(o), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int j = 0; j < 4; ++j) printf(" %4d", o[l * 4 + j]); printf("\n"); }
  return 0;
}
