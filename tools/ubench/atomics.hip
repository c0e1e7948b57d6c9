// atomics.hip -- what do order-independent BatchNorm statistics cost?  
// A producer conv has 256 workgroups, each with one row of per-channel partial sums (sum, sum of squares).  Today they are
// stored as rows and a single-workgroup launch (bn_finalize) reduces them.  The alternative: every workgroup ADDS its row
// to 2 C words with device-scope int64 atomics (integer addition is associative: bitwise deterministic), the consumer reads
// 2 C words.  This probe measures the tail such atomics add to a kernel:
//   grid G workgroups x 256 threads; thread t < W adds one value to word ((t + rot * blockIdx) % W) * stride
//   variants: int64 add / fp64 add / plain store of a row (today's form) / nothing
// and, as the consumer side, the cost of reading W words in a prologue (not measured here: it is one L2 round trip).
// build: hipcc --offload-arch=gfx950 -O3 -o atomics atomics.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// MODE 0: nothing, 1: int64 atomic add (no return), 2: fp64 atomic add, 3: row store (one row per workgroup), 4: int64 atomics issued by
// workgroups of one XCD leader only after an LDS reduce (not applicable here) -- unused
template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned long long* acc, double* accd, float* rows, int W, int stride, int rot, int work) {
  const int t = threadIdx.x;
  // some real work in front so that the atomics are a tail, not the whole kernel: `work` dependent FMAs
  float v = (float)(t + blockIdx.x);
  for (int i = 0; i < work; ++i) v = fmaf(v, 1.0001f, 0.5f);
  if (MODE == 0) {
    if (v == 123.456f) rows[0] = v;
    return;
  }
  for (int w = t; w < W; w += 256) {
    const int idx = ((w + rot * (int)blockIdx.x) % W) * stride;
    if (MODE == 1) {
      const long long q = (long long)(v * 1048576.0f);
      __hip_atomic_fetch_add(reinterpret_cast<long long*>(acc) + idx, q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (MODE == 2) {
      __hip_atomic_fetch_add(accd + idx, (double)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (MODE == 3) {
      rows[(long long)blockIdx.x * W + w] = v;
    }
  }
}

// the single-workgroup reduction that MODE 3 needs afterwards (bn_finalize's shape: 1024 threads, fp64 sums over the rows)
__global__ __launch_bounds__(1024) void finalize(const float* rows, float* out, int G, int W) {
  __shared__ double red[8][128];
  for (int c0 = 0; c0 < W; c0 += 128) {
    const int cl = threadIdx.x & 127, rg = threadIdx.x >> 7;
    double s = 0.0;
    if (c0 + cl < W)
      for (int r = rg; r < G; r += 8) s += rows[(long long)r * W + c0 + cl];
    red[rg][cl] = s;
    __syncthreads();
    if (threadIdx.x < 128 && c0 + threadIdx.x < W) {
      double tsum = 0.0;
      for (int g = 0; g < 8; ++g) tsum += red[g][threadIdx.x];
      out[c0 + threadIdx.x] = (float)tsum;
    }
    __syncthreads();
  }
}

// a consumer-like streaming kernel to put between launches (so that the chain looks like the network: wide kernel, glue, wide kernel)
__global__ __launch_bounds__(256) void stream(const float4* src, float4* dst, long long n) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += gridDim.x * 256ll) dst[i] = src[i];
}

template <int MODE>
float run(int G, int W, int stride, int rot, int work, int reps, unsigned long long* acc, double* accd, float* rows, float* out, bool fin,
          const float4* src, float4* dst, long long nstream) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int warm = 0; warm < 2; ++warm) {
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) {
      if (nstream) stream<<<2048, 256>>>(src, dst, nstream);
      k<MODE><<<G, 256>>>(acc, accd, rows, W, stride, rot, work);
      if (fin) finalize<<<1, 1024>>>(rows, out, G, W);
    }
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
  }
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1000.f / reps;
}

int main() {
  const int G = 256;
  unsigned long long* acc; double* accd; float *rows, *out; float4 *src, *dst;
  const size_t ACC = 4096ull * 512 * 8;
  CK(hipMalloc(&acc, ACC)); CK(hipMalloc(&accd, ACC)); CK(hipMalloc(&rows, 2048ull * 4096 * 4)); CK(hipMalloc(&out, 4096 * 4));
  const long long nstream = 8ll << 20;   // 128 MB copy between launches
  CK(hipMalloc(&src, nstream * 16)); CK(hipMalloc(&dst, nstream * 16));
  CK(hipMemset(acc, 0, ACC)); CK(hipMemset(accd, 0, ACC)); CK(hipMemset(src, 0, nstream * 16));
  const int reps = 200;
  printf("# per-launch time in us (launches back to back on one stream, %d reps); G = workgroups, W = words per workgroup\n", reps);
  for (int withstream = 0; withstream < 2; ++withstream) {
    const long long ns = withstream ? nstream : 0;
    printf("## %s\n", withstream ? "each probe launch behind a 128 MB streaming copy (2048 workgroups)" : "probe launches only");
    for (int Gx : {256, 1024}) {
      for (int W : {64, 256, 1024}) {
        const float t0 = run<0>(Gx, W, 1, 0, 2000, reps, acc, accd, rows, out, false, src, dst, ns);
        const float t3 = run<3>(Gx, W, 1, 0, 2000, reps, acc, accd, rows, out, false, src, dst, ns);
        const float t3f = run<3>(Gx, W, 1, 0, 2000, reps, acc, accd, rows, out, true, src, dst, ns);
        printf("G %4d W %4d  nothing %7.2f  row store %7.2f  row store + finalize launch %7.2f\n", Gx, W, t0, t3, t3f);
        for (int stride : {1, 8, 32, 512}) {
          for (int rot : {0, 1, 7}) {
            const float t1 = run<1>(Gx, W, stride, rot, 2000, reps, acc, accd, rows, out, false, src, dst, ns);
            const float t2 = run<2>(Gx, W, stride, rot, 2000, reps, acc, accd, rows, out, false, src, dst, ns);
            printf("G %4d W %4d  stride %4d words rot %d   int64 atomics %7.2f (+%6.2f)   fp64 atomics %7.2f (+%6.2f)\n", Gx, W, stride, rot, t1, t1 - t0,
                   t2, t2 - t0);
          }
        }
      }
    }
  }
  // correctness of the int64 sum: every workgroup adds 1 to W words -> each word == G * launches
  CK(hipMemset(acc, 0, ACC));
  return 0;
}
