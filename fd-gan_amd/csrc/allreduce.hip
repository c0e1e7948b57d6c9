// allreduce.hip -- fdgan_allreduce_*: the data-parallel gradient exchange behind the C ABI, for a host that is not PyTorch
// (SURVEY 8(b) lists it; the reference's own mechanism is nn.DataParallel, /root/reference/demo.py:89).  The Python host of this
// repository does NOT use it: fdgan_hip/optim.py reduces the same two flat gradient buffers with torch.distributed (RCCL on its
// own stream, overlapped with the backward walk), because the framework already owns the communicator and the rendezvous.
// What is here is the thinnest possible wrapper: RCCL is looked up at run time (dlopen -- the library has no link-time
// dependency on it and loads on a box without RCCL), the caller distributes the 128-byte unique id out of band, and the sum runs
// in place on the caller's stream.  One process per GPU, one communicator per process.
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>

#include <mutex>

#include "common.h"

namespace {

struct RcclId {
  char bytes[128];   // ncclUniqueId (NCCL_UNIQUE_ID_BYTES)
};
typedef int (*GetUniqueIdFn)(RcclId*);
typedef int (*CommInitRankFn)(void**, int, RcclId, int);
typedef int (*AllReduceFn)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*CommDestroyFn)(void*);
typedef const char* (*GetErrorStringFn)(int);

struct Rccl {
  void* handle = nullptr;
  GetUniqueIdFn get_unique_id = nullptr;
  CommInitRankFn comm_init_rank = nullptr;
  AllReduceFn all_reduce = nullptr;
  CommDestroyFn comm_destroy = nullptr;
  GetErrorStringFn error_string = nullptr;
};

int rccl_load(Rccl** out) {
  static Rccl lib;
  static int state = 0;   // 1 ok, -1 failed
  static char why[256] = "symbols missing";
  static std::once_flag once;
  std::call_once(once, [] {
    // a process that already runs torch has its RCCL loaded: the soname resolves to that copy
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
      lib.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (lib.handle) break;
      if (const char* e = dlerror()) snprintf(why, sizeof(why), "%s", e);   // dlerror() clears itself: read it once, right here
    }
    if (lib.handle) {
      lib.get_unique_id = reinterpret_cast<GetUniqueIdFn>(dlsym(lib.handle, "ncclGetUniqueId"));
      lib.comm_init_rank = reinterpret_cast<CommInitRankFn>(dlsym(lib.handle, "ncclCommInitRank"));
      lib.all_reduce = reinterpret_cast<AllReduceFn>(dlsym(lib.handle, "ncclAllReduce"));
      lib.comm_destroy = reinterpret_cast<CommDestroyFn>(dlsym(lib.handle, "ncclCommDestroy"));
      lib.error_string = reinterpret_cast<GetErrorStringFn>(dlsym(lib.handle, "ncclGetErrorString"));
    }
    if (lib.handle && !(lib.get_unique_id && lib.comm_init_rank && lib.all_reduce && lib.comm_destroy))
      snprintf(why, sizeof(why), "symbols missing");
    state = (lib.handle && lib.get_unique_id && lib.comm_init_rank && lib.all_reduce && lib.comm_destroy) ? 1 : -1;
  });
  if (state != 1) FD_FAIL(FD_EUNSUPPORTED, "allreduce: RCCL (librccl.so) is not available in this process: %s", why);
  *out = &lib;
  return FD_OK;
}

#define RCCL_CHECK(lib, call, what)                                                                                   \
  do {                                                                                                                \
    const int rc_ = (call);                                                                                           \
    if (rc_ != 0) FD_FAIL(FD_ELAUNCH, "%s: RCCL error %d (%s)", what, rc_, (lib)->error_string ? (lib)->error_string(rc_) : "?"); \
  } while (0)

}  // namespace

extern "C" int fdgan_allreduce_unique_id(void* id128) {
  FD_REQUIRE(id128, "allreduce_unique_id: NULL pointer");
  Rccl* lib;
  if (int rc = rccl_load(&lib)) return rc;
  RcclId id;
  RCCL_CHECK(lib, lib->get_unique_id(&id), "allreduce_unique_id");
  memcpy(id128, id.bytes, sizeof(id.bytes));
  return FD_OK;
}

extern "C" int fdgan_allreduce_comm_create(const void* id128, int rank, int world, void** comm) {
  FD_REQUIRE(id128 && comm && world >= 1 && rank >= 0 && rank < world, "allreduce_comm_create: bad arguments (rank %d of %d)", rank, world);
  Rccl* lib;
  if (int rc = rccl_load(&lib)) return rc;
  RcclId id;
  memcpy(id.bytes, id128, sizeof(id.bytes));
  *comm = nullptr;
  RCCL_CHECK(lib, lib->comm_init_rank(comm, world, id, rank), "allreduce_comm_create");
  return FD_OK;
}

extern "C" int fdgan_allreduce_sum_f32(void* comm, float* buf, int64_t count, FdStream stream) {
  FD_REQUIRE(comm && buf && count > 0, "allreduce_sum_f32: bad arguments");
  Rccl* lib;
  if (int rc = rccl_load(&lib)) return rc;
  // ncclFloat32 = 7, ncclSum = 0 (nccl.h); in place, asynchronous on the caller's stream
  RCCL_CHECK(lib, lib->all_reduce(buf, buf, (size_t)count, 7, 0, comm, static_cast<hipStream_t>(stream)), "allreduce_sum_f32");
  return FD_OK;
}

extern "C" int fdgan_allreduce_comm_destroy(void* comm) {
  FD_REQUIRE(comm, "allreduce_comm_destroy: NULL communicator");
  Rccl* lib;
  if (int rc = rccl_load(&lib)) return rc;
  RCCL_CHECK(lib, lib->comm_destroy(comm), "allreduce_comm_destroy");
  return FD_OK;
}
