// Shared host-side plumbing of libfdgan_hip.so: error reporting, the launch
// recorder behind the FdPlan API, and small device helpers.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <utility>
#include <vector>

#include "../../include/fdgan_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) float f32x8;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

// ---- the two 16-bit element formats ------------------------------------------------
// FmtA ("activation"): everything the FORWARD pass stores or multiplies -- activations, the forward filter images -- is
//   IEEE fp16 (11-bit significand).  Measured on the CPU oracle (tools/precision_study.py): with bf16 storage the
//   generator's output sits 46.8 dB from the fp32 reference whatever the kernels do, with fp16 at 64.4 dB; the
//   north-star's 0.02 dB metric budget needs >= 54 dB.  Normalised activations never come near fp16's 65504.
// FmtG ("gradient"): activation gradients and the flipped filter images the data-gradient kernels multiply them with
//   stay bf16 -- unscaled loss gradients (1e-7 per pixel for a mean over 16 x 3 x 256 x 256) need fp32's exponent range.
// Both are 2 bytes: layouts, strides and byte counts are identical; v_mfma_f32_16x16x32_{f16,bf16} run at the same rate.
struct FmtA {
  typedef f16x8 v8;
  typedef f16x4 v4;
  typedef f16x2 v2;
  typedef _Float16 T;
  static constexpr int DT = FD_F16;
};
struct FmtG {
  typedef bf16x8 v8;
  typedef bf16x4 v4;
  typedef bf16x2 v2;
  typedef __bf16 T;
  static constexpr int DT = FD_BF16;
};
template <bool GRAD> struct FmtSel { typedef FmtA type; };
template <> struct FmtSel<true> { typedef FmtG type; };

template <class F> __device__ __forceinline__ f32x8 fd_cvt8(u32x4 raw) {
  return __builtin_convertvector(__builtin_bit_cast(typename F::v8, raw), f32x8);
}
template <class F> __device__ __forceinline__ f32x4 fd_cvt4(u32x2 raw) {
  return __builtin_convertvector(__builtin_bit_cast(typename F::v4, raw), f32x4);
}
template <class F> __device__ __forceinline__ f32x2 fd_cvt2(unsigned raw) {
  return __builtin_convertvector(__builtin_bit_cast(typename F::v2, raw), f32x2);
}
template <class F> __device__ __forceinline__ float fd_cvt1(unsigned short raw) {
  return (float)__builtin_bit_cast(typename F::T, raw);
}
template <class F> __device__ __forceinline__ u32x4 fd_pk8(f32x8 f) {
  return __builtin_bit_cast(u32x4, __builtin_convertvector(f, typename F::v8));
}
template <class F> __device__ __forceinline__ u32x2 fd_pk4(f32x4 f) {
  return __builtin_bit_cast(u32x2, __builtin_convertvector(f, typename F::v4));
}
template <class F> __device__ __forceinline__ unsigned fd_pk2(f32x2 f) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f, typename F::v2));
}
template <class F> __device__ __forceinline__ unsigned short fd_pk1(float f) {
  return __builtin_bit_cast(unsigned short, (typename F::T)f);
}
// ---- stochastic rounding for ACCUMULATIONS into bf16 gradient buffers ---------------------------------------------------------
// A gradient buffer holds bf16 values, i.e. numbers ON the bf16 grid.  Adding a small term d to such a value g and rounding to
// nearest returns g whenever |d| < ulp(g) / 2: the term is not rounded, it is ABSORBED -- every time, for every pixel, with the
// same sign.  BatchNorm's backward adds exactly such terms: dx = A dpre + (B x + C), where B x + C (the subtraction of the
// batch mean of the gradient) is ~1 / sqrt(N H W) of A dpre and is added in a second pass (the deferred affine of
// fdgan_hip/backward.py).  Round 4 measured what that does (tools/dbg_dc.py): the gradient a BatchNorm backward leaves sums to
// 3e-4 .. 8e-4 of its absolute sum per channel instead of zero (round-to-nearest noise would be 1e-5), and a weight gradient
// taken against activations with a large mean -- the well-conditioned fixtures, BatchNorm biases + 3 -- is 5-20 % off, as a
// rank-1 error (DC offset x sum of the input).  Rounding the SUM stochastically (v_cvt_sr_bf16_f32: up with probability = the
// discarded fraction) is unbiased whatever the sizes of the two terms, costs the same 16 bits, and stays reproducible: the
// random bits are a hash of the element's position, not of time.
__device__ __forceinline__ unsigned fd_mix32(unsigned x) {
  x ^= x >> 16;
  x *= 0x7feb352du;
  x ^= x >> 15;
  x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}
// 8 fp32 -> 8 bf16 (4 dwords), each rounded stochastically; `seed`: any number unique to this 8-element group in the launch
__device__ __forceinline__ u32x4 fd_pk8_sr(f32x8 f, unsigned seed) {
  const unsigned h = fd_mix32(seed);
  u32x4 out;
#pragma unroll
  for (int k = 0; k < 4; ++k) {   // a fresh 16 random bits per element: rotations of one well-mixed word (v_alignbit_b32)
    bf16x2 t = {0, 0};
    t = __builtin_amdgcn_cvt_sr_bf16_f32(t, f[2 * k], __builtin_amdgcn_alignbit(h, h, 8 * k + 3), false);
    t = __builtin_amdgcn_cvt_sr_bf16_f32(t, f[2 * k + 1], __builtin_amdgcn_alignbit(h, h, 8 * k + 19), true);
    out[k] = __builtin_bit_cast(unsigned, t);
  }
  return out;
}
__device__ __forceinline__ unsigned short fd_pk1_sr(float f, unsigned seed) {
  bf16x2 t = {0, 0};
  t = __builtin_amdgcn_cvt_sr_bf16_f32(t, f, fd_mix32(seed), false);
  return (unsigned short)(__builtin_bit_cast(unsigned, t) & 0xffffu);
}
// Where it is used, and where NOT (measured, round 4): only where the added term is small AND coherent -- fdgan_affine_accumulate and
// the dy staging of the fused bottleneck backward (both add B x + C), the four-scale head's input gradient (constants over
// k x k windows).  In the data-gradient kernels' own `G += gamma * rstd * dpre` the two terms are of one size, nearest rounding has
// no bias to remove, and stochastic rounding only doubles the variance: all 282 generator gradients moved from a median 0.80 % /
// p90 1.20 % to 0.89 % / 1.45 % from the reference's with it there (and the dominant kernel lost 5 %), so those stores round to nearest.

// D = A x B + C on one 16 x 16 x 32 tile; operands as raw 16-byte fragments
template <class F> __device__ __forceinline__ f32x4 fd_mfma(u32x4 a, u32x4 b, f32x4 c);
template <> __device__ __forceinline__ f32x4 fd_mfma<FmtA>(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x4 fd_mfma<FmtG>(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 fd_mfma_a(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 fd_mfma_g(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

// ---- tuning switches ------------------------------------------------------------
// Kernel-selection / phase-skipping switches used by tools/ while tuning.  They exist only in builds made with
// -DFDGAN_TUNING (FDGAN_TUNING=1 python __graft_entry__.py); in the shipped library the macro is a constant NULL, the
// branches fold away and the switch names do not appear in the binary (tests/test_capi_cpu.py checks).
#ifdef FDGAN_TUNING
#include <stdlib.h>
#define FD_TUNE_GETENV(name) getenv(name)
#else
#define FD_TUNE_GETENV(name) (static_cast<const char*>(nullptr))
#endif

// ---- errors ---------------------------------------------------------------
void fd_set_error(const char* fmt, ...);
// CU budget of the launches being issued (fdgan_set_cu_budget, include/fdgan_hip.h): the persistent kernels -- one (or two) resident
// workgroups per CU walking the work -- size their grids by it instead of the device's CU count, so that a stream created with a CU
// mask (hipExtStreamCreateWithCUMask) gets a grid that fills exactly its share of the chip.  0: no budget, the launcher's default.
int fd_cu_budget();
static inline int fd_cus(int dflt) {
  const int b = fd_cu_budget();
  return b > 0 && b < dflt ? b : dflt;
}

#define FD_FAIL(code, ...)      \
  do {                          \
    fd_set_error(__VA_ARGS__);  \
    return (code);              \
  } while (0)
#define FD_REQUIRE(cond, ...) \
  do {                        \
    if (!(cond)) FD_FAIL(FD_EINVAL, __VA_ARGS__); \
  } while (0)

// ---- launch recorder --------------------------------------------------------
// Every kernel takes ONE by-value POD argument struct, so a launch is fully
// described by (function, grid, block, dynamic LDS, bytes of the struct).
struct FdLaunch {
  const void* fn;      // nullptr: not a kernel but a stream dependency -- stream `slot` waits for everything enqueued so far on stream `other`
  const char* name;
  dim3 grid, block;
  unsigned shmem;
  std::vector<char> arg;
  int slot = 0;        // which of the streams handed to fdgan_plan_launch_multi this launch goes to (0: the launch stream)
  int other = 0;
};

struct FdPlan {
  std::vector<FdLaunch> launches;
  bool recording = false;
  int cur_slot = 0;                                       // stream slot of the launches being recorded (fdgan_plan_set_slot)
  int max_slot = 0;
  std::vector<hipEvent_t> wait_events;                    // one per recorded dependency, created by the first multi-stream launch
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  std::vector<int64_t> marked;                            // launch indices bracketed by events
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pairs;  // recorded, not yet read
};

// Enqueue on `stream`, or append to the plan being recorded by this thread.
int fd_enqueue(const void* fn, const char* name, dim3 grid, dim3 block, unsigned shmem,
               const void* arg, size_t arg_bytes, hipStream_t stream);

template <typename Args>
static inline int fd_launch(void (*kernel)(Args), const char* name, dim3 grid, dim3 block,
                            unsigned shmem, const Args& a, hipStream_t stream) {
  return fd_enqueue(reinterpret_cast<const void*>(kernel), name, grid, block, shmem, &a, sizeof(Args), stream);
}

// ---- device helpers -----------------------------------------------------------
__device__ __forceinline__ float fd_act(float v, int act) {
  switch (act) {
    case FD_ACT_RELU: return fmaxf(v, 0.f);
    case FD_ACT_LEAKY02: return fmaxf(v, 0.2f * v);
    case FD_ACT_TANH: return tanhf(v);
    case FD_ACT_SIGMOID: return 1.f / (1.f + __expf(-v));
    default: return v;
  }
}
