// gridbar.hip -- dense block 3 as ONE persistent launch with grid barriers vs the same phases as separate launches (experiment
// aid).     hipcc --offload-arch=gfx950 -O3 -w -o /tmp/gridbar tools/ubench/gridbar.hip && /tmp/gridbar
//
// The question: four dense layers at 64 x 64 (B = 16: 65 536 pixels) are 16 dependent launches in the forward pass -- 1x1 conv
// (reads the C-channel prefix, writes the 128-channel bottleneck + statistics partials), statistics finalize (one small
// workgroup's worth of work), 3x3 conv (reads the bottleneck, writes 32 channels + partials), finalize -- each 5-40 us, and the
// block's working set (134 MB) sits in the 256 MB Infinity Cache either way.  Would one cooperative launch that walks the same
// phases behind an XCD-hierarchical grid barrier be faster than the launches?  No data stays on chip between the phases (the 3x3's
// halo comes from neighbouring workgroups through L2 / MALL), so what a persistent kernel can save is the launch boundary, and
// what it pays is the barrier.  This benchmark isolates exactly that: the phases are streaming passes with the dense layers' BYTE
// COUNTS (no MFMA work: both variants would do the same), run (a) as separate launches on one stream, (b) as one launch of one
// workgroup per CU with the barrier of /opt/skills/guides/MI355X_MICROARCH.md ("barrier-xcd": per-XCC counter, XCD leader release
// fence -> top counter -> per-XCC generation word, every workgroup an agent-scope acquire).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#define NCU 256
#define NT 512

struct Phase {
  long long rd16, wr16;   // 16-byte pieces read / written by the whole grid
  int small;              // 1: a finalize -- only the first 4 workgroups have work
};
#define MAXPH 64
struct Args {
  const u32x4* src;
  u32x4* dst;
  unsigned* bar;          // [0..7] per-XCC arrival counters (128 B apart), [8] top counter, [9..16] per-XCC generation, [17] census done
  int nph;
  Phase ph[MAXPH];
};

__device__ __forceinline__ void phase_body(const Args& a, const Phase& p, int wg, int nwg, int tid) {
  const int active = p.small ? (nwg < 4 ? nwg : 4) : nwg;
  if (wg >= active) return;
  const long long stride = (long long)active * NT;
  u32x4 acc = {0u, 0u, 0u, 0u};
  long long i = (long long)wg * NT + tid;
  for (; i + 3 * stride < p.rd16; i += 4 * stride) {
    const u32x4 v0 = a.src[i], v1 = a.src[i + stride], v2 = a.src[i + 2 * stride], v3 = a.src[i + 3 * stride];
    acc += v0 ^ v1 ^ v2 ^ v3;
  }
  for (; i < p.rd16; i += stride) acc += a.src[i];
  for (long long o = (long long)wg * NT + tid; o < p.wr16; o += stride) a.dst[o] = acc;
}

__global__ __launch_bounds__(NT) void phase_kernel(Args a, int k) {
  extern __shared__ char pad[];   // 100 KB: one workgroup per CU, as the dense-layer kernels
  phase_body(a, a.ph[k], blockIdx.x, gridDim.x, threadIdx.x);
}

__device__ __forceinline__ unsigned ld_relaxed(unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// XCD-hierarchical barrier, monotonic counters; gen = 1, 2, ...; per_xcc[x] = workgroups resident on XCC x (census of phase 0)
__device__ __forceinline__ void grid_barrier(unsigned* bar, unsigned gen, int xcc, unsigned mine, int nxcc_used) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned arrived = __hip_atomic_fetch_add(bar + xcc * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
    if (arrived == mine * gen) {   // this XCD's last arriver: its leader
      const unsigned t = __hip_atomic_fetch_add(bar + 8 * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
      if (t == (unsigned)nxcc_used * gen)
        for (int x = 0; x < 8; ++x) __hip_atomic_store(bar + (9 + x) * 32, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    unsigned spins = 0;
    while (ld_relaxed(bar + (9 + xcc) * 32) < gen && ++spins < (1u << 18)) __builtin_amdgcn_s_sleep(2);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}

__global__ __launch_bounds__(NT) void persistent_kernel(Args a, unsigned launch_idx) {
  extern __shared__ char pad[];
  __shared__ unsigned s_mine, s_used;
  unsigned* bar = a.bar + (size_t)launch_idx * 32 * 32;   // every launch its own zeroed set of words (128 B apart)
  const int xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 7;
  // census: how many workgroups sit on my XCC (HIP promises no placement): one flat barrier on a spare counter set
  unsigned* cen = bar + 20 * 32;
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(cen + xcc * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(cen + 8 * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (ld_relaxed(cen + 8 * 32) < gridDim.x && ++spins < (1u << 18)) __builtin_amdgcn_s_sleep(2);
    s_mine = ld_relaxed(cen + xcc * 32);
    unsigned used = 0;
    for (int x = 0; x < 8; ++x) used += ld_relaxed(cen + x * 32) != 0;
    s_used = used;
  }
  __syncthreads();
  const unsigned mine = s_mine, used = s_used;
  unsigned gen = 0;
  for (int k = 0; k < a.nph; ++k) {
    phase_body(a, a.ph[k], blockIdx.x, gridDim.x, threadIdx.x);
    if (k + 1 < a.nph) grid_barrier(bar, ++gen, xcc, mine, (int)used);
  }
}

int main(int argc, char** argv) {
  const long long P = 16ll * 64 * 64;
  Args a;
  memset(&a, 0, sizeof(a));
  const size_t bytes = 256ull << 20;
  hipMalloc((void**)&a.src, bytes), hipMalloc((void**)&a.dst, bytes), hipMalloc((void**)&a.bar, 128 * 4096);
  hipMemset((void*)a.src, 1, bytes), hipMemset(a.bar, 0, 128 * 4096);
  const int layers = argc > 1 ? atoi(argv[1]) : 4, c0 = argc > 2 ? atoi(argv[2]) : 576;   // dense block 3, layers 11-14: C = 576 ..
  int n = 0;
  double tot_bytes = 0;
  for (int l = 0; l < layers; ++l) {
    const long long c = c0 + 32 * l;
    a.ph[n++] = Phase{P * c * 2 / 16, P * 128 * 2 / 16 + NCU * 256 * 4 / 16, 0};   // 1x1: prefix in, bottleneck + partials out
    a.ph[n++] = Phase{NCU * 256 * 4 / 16, 128 * 8 / 16, 1};                         // finalize
    a.ph[n++] = Phase{P * 128 * 2 / 16, P * 32 * 2 / 16 + NCU * 64 * 4 / 16, 0};    // 3x3: bottleneck in, 32 channels + partials out
    a.ph[n++] = Phase{NCU * 64 * 4 / 16, 32 * 8 / 16, 1};                           // finalize
    tot_bytes += (double)P * (c + 128 + 128 + 32) * 2;
  }
  a.nph = n;
  hipFuncSetAttribute((const void*)phase_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipFuncSetAttribute((const void*)persistent_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  const int reps = 20;
  float ms_l = 1e9f, ms_p = 1e9f, ms_1 = 1e9f;
  for (int round = 0; round < 4; ++round) {
    hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r)
      for (int k = 0; k < n; ++k) hipLaunchKernelGGL(phase_kernel, dim3(NCU), dim3(NT), 100 * 1024, 0, a, k);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (round && ms / reps < ms_l) ms_l = ms / reps;
  }
  unsigned launches = 0;
  for (int round = 0; round < 4; ++round) {
    hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r, ++launches) hipLaunchKernelGGL(persistent_kernel, dim3(NCU), dim3(NT), 100 * 1024, 0, a, launches);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (round && ms / reps < ms_p) ms_p = ms / reps;
  }
  // the phases' own time: every phase alone, repeated back to back (no dependence on a different kernel's tail)
  double sum_alone = 0;
  for (int k = 0; k < n; ++k) {
    float best = 1e9f;
    for (int round = 0; round < 3; ++round) {
      hipEventRecord(e0, 0);
      for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(phase_kernel, dim3(NCU), dim3(NT), 100 * 1024, 0, a, k);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (ms / reps < best) best = ms / reps;
    }
    sum_alone += best;
  }
  (void)ms_1;
  unsigned timeout_probe[32 * 32];
  hipMemcpy(timeout_probe, a.bar + 79 * 32 * 32, sizeof(timeout_probe), hipMemcpyDeviceToHost);   // the last launch's words
  printf("dense block 3 stand-in: %d layers from C = %d at 16 x 64 x 64, %d phases, %.1f MB per pass\n", layers, c0, n, tot_bytes / 1e6);
  printf("  (a) %2d separate launches            : %8.1f us per pass   (%.2f TB/s)\n", n, ms_l * 1e3, tot_bytes / ms_l / 1e9);
  printf("  (b) one persistent launch, %2d barriers: %8.1f us per pass   (%.2f TB/s)\n", n - 1, ms_p * 1e3, tot_bytes / ms_p / 1e9);
  printf("      sum of the phases launched alone  : %8.1f us  -> boundary cost (a) %.2f us per launch, barrier cost (b) %.2f us each\n",
         sum_alone * 1e3, (ms_l - sum_alone) * 1e3 / n, (ms_p - sum_alone) * 1e3 / (n - 1));
  printf("  persistent / launches = %.3f  (adopt below 0.85)\n", ms_p / ms_l);
  printf("  last launch's barrier words (expect top = 8 x %d, arrivals = 32 x %d each): top %u, per-XCC arrivals %u %u %u %u %u %u %u %u\n", n - 1, n - 1, timeout_probe[8 * 32], timeout_probe[0], timeout_probe[32], timeout_probe[64],
         timeout_probe[96], timeout_probe[128], timeout_probe[160], timeout_probe[192], timeout_probe[224]);
  return 0;
}
