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
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) float f32x8;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

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
  const void* fn;
  const char* name;
  dim3 grid, block;
  unsigned shmem;
  std::vector<char> arg;
};

struct FdPlan {
  std::vector<FdLaunch> launches;
  bool recording = false;
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
