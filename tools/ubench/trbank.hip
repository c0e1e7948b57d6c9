// trbank.hip -- LDS bank behaviour of ds_read_b64_tr_b16 / ds_read_b128 / ds_write_b128 for the address patterns the
// weight-gradient kernels use.  Each pattern is a table of 64 per-lane byte offsets; the kernel issues N reads per wave
// (8 waves per workgroup, one workgroup per CU) and the host reports LDS cycles per wave-instruction per CU.
// build: hipcc --offload-arch=gfx950 -O3 -o trbank trbank.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include <string>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>   // 0: tr read b64, 1: read b128, 2: write b128, 3: plain read b64
__global__ __launch_bounds__(512) void k(const int* offs, int iters, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 160 * 1024 / 4 - 64; i += blockDim.x) reinterpret_cast<unsigned*>(lds)[i] = i * 2654435761u;
  __syncthreads();
  const int base = offs[lane] + wave * 32;   // each wave its own 32-byte channel group (as the kernels do)
  unsigned acc = 0;
  u32x4 wv = {1u, 2u, 3u, (unsigned)lane};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const char* p = lds + base + (u & 7) * 8192 + (u >> 3) * 4096;
      if constexpr (MODE == 0) {
        s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
        acc ^= (unsigned)v[0] ^ (unsigned)v[3];
      } else if constexpr (MODE == 1) {
        u32x4 v = *reinterpret_cast<const u32x4*>(p);
        acc ^= v[0] ^ v[3];
      } else if constexpr (MODE == 3) {
        const unsigned long long v = *reinterpret_cast<const unsigned long long*>(p);
        acc ^= (unsigned)v ^ (unsigned)(v >> 32);
      } else {
        *reinterpret_cast<u32x4*>(const_cast<char*>(p)) = wv;
      }
    }
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

struct Pat { std::string name; int mode; int offs[64]; };

int main() {
  std::vector<Pat> pats;
  auto add = [&](const char* name, int mode, auto f) { Pat p; p.name = name; p.mode = mode; for (int l = 0; l < 64; ++l) p.offs[l] = f(l); pats.push_back(p); };
  // tr-read lane map of the kernels: g = l >> 4, i = l & 15, pixel 8 g + (i >> 2), piece (i & 3) * 8
  auto trpix = [](int l) { return 8 * (l >> 4) + ((l & 15) >> 2); };
  auto piece = [](int l) { return (l & 3) * 8; };
  add("linear 8B (ideal)", 0, [&](int l) { return l * 8; });
  add("linear 8B plain b64", 3, [&](int l) { return l * 8; });
  for (int kx = 0; kx < 3; ++kx) {
    char nm[64]; snprintf(nm, 64, "tr8 x-row xor swizzle, tap shift %d", kx);
    add(nm, 0, [&](int l) { int pix = trpix(l) + kx; return pix * 256 + ((0 ^ ((pix & 3) | (((pix >> 3) & 1) << 2))) << 5) + piece(l); });
  }
  add("tr8 x-row no swizzle", 0, [&](int l) { int pix = trpix(l); return pix * 256 + piece(l); });
  for (int kx = 0; kx < 3; ++kx) {
    char nm[64]; snprintf(nm, 64, "tr8 second half (+4 px), shift %d", kx);
    add(nm, 0, [&](int l) { int pix = trpix(l) + 4 + kx; return pix * 256 + ((0 ^ ((pix & 3) | (((pix >> 3) & 1) << 2))) << 5) + piece(l); });
  }
  add("dy row (64 B/pixel, xor)", 0, [&](int l) { int pix = trpix(l); return pix * 64 + ((0 ^ ((pix >> 3) & 1)) << 5) + piece(l); });
  for (int kx = 0; kx < 4; ++kx) {
    char nm[64]; snprintf(nm, 64, "tr9 x-row (288 B + 128/8px), shift %d", kx);
    add(nm, 0, [&](int l) { int pix = trpix(l) + kx; return pix * 288 + (pix >> 3) * 128 + piece(l); });
  }
  for (int kx = 0; kx < 3; ++kx) {
    char nm[64]; snprintf(nm, 64, "tr3 x-row (96 B + 128/8px), shift %d", kx);
    add(nm, 0, [&](int l) { int pix = trpix(l) + kx; return pix * 96 + (pix >> 3) * 128 + piece(l); });
  }
  // candidate layouts: pixel pitch 256 + 8 B / 256 + 16 / 256 + 32 (padding instead of xor)
  for (int pad : {8, 16, 32, 64}) {
    char nm[64]; snprintf(nm, 64, "pitch 256+%d", pad);
    add(nm, 0, [&](int l) { int pix = trpix(l); return pix * (256 + pad) + piece(l); });
  }
  add("write b128 x-row (16 lanes/pixel)", 2, [&](int l) { int pix = l >> 4, ch = l & 15; return pix * 256 + (((ch >> 1) ^ ((pix & 3) | (((pix >> 3) & 1) << 2))) << 5) + ((ch & 1) << 4); });
  add("read b128 linear", 1, [&](int l) { return l * 16; });

  int* doffs; unsigned* sink;
  hipMalloc(&doffs, 256); hipMalloc(&sink, 16);
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const double ghz = prop.clockRate * 1e-6;
  printf("clock %.2f GHz, %d CUs\n", ghz, prop.multiProcessorCount);
  const int iters = 2000, waves = 8;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (auto& p : pats) {
    hipMemcpy(doffs, p.offs, 256, hipMemcpyHostToDevice);
    auto launch = [&]() {
      const dim3 g(prop.multiProcessorCount), b(64 * waves);
      if (p.mode == 0) { hipFuncSetAttribute((const void*)&k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); k<0><<<g, b, 160 * 1024 - 256>>>(doffs, iters, sink); }
      if (p.mode == 1) { hipFuncSetAttribute((const void*)&k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); k<1><<<g, b, 160 * 1024 - 256>>>(doffs, iters, sink); }
      if (p.mode == 2) { hipFuncSetAttribute((const void*)&k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); k<2><<<g, b, 160 * 1024 - 256>>>(doffs, iters, sink); }
      if (p.mode == 3) { hipFuncSetAttribute((const void*)&k<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); k<3><<<g, b, 160 * 1024 - 256>>>(doffs, iters, sink); }
    };
    launch(); hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 3; ++r) { hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
    const double insts = (double)iters * 16 * waves;   // wave-instructions per CU
    printf("%-42s %8.3f ms  %6.2f cycles / wave-instruction / CU\n", p.name.c_str(), best, best * 1e-3 * ghz * 1e9 / insts);
  }
  hipError_t e = hipGetLastError();
  printf("status: %s\n", hipGetErrorString(e));
  return 0;
}
