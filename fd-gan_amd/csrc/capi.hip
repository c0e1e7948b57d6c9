// capi.hip -- error state, the launch recorder (FdPlan) and library identity.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "common.h"

static thread_local char g_err[512] = "";
static thread_local FdPlan* g_recording = nullptr;

void fd_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* fdgan_last_error(void) { return g_err; }
extern "C" int fdgan_version(void) { return FDGAN_ABI_VERSION; }

static thread_local int g_cu_budget = 0;
int fd_cu_budget() { return g_cu_budget; }
extern "C" int fdgan_set_cu_budget(int ncu) {
  const int prev = g_cu_budget;
  g_cu_budget = ncu > 0 ? ncu : 0;
  return prev;
}

extern "C" const char* fdgan_device_arch(void) {
  static thread_local char arch[256];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return nullptr;
  strncpy(arch, prop.gcnArchName, sizeof(arch) - 1);
  arch[sizeof(arch) - 1] = 0;
  return arch;
}

// ---- kernel timer: hipEvent pairs around selected launches, on the stream they are launched on -----------------
// (bench.py's roofline leg: the average duration of the dominant kernel inside the timed region; every launch of
// the library -- eager or replayed from a plan -- passes through do_launch)
namespace {
struct KernelTimer {
  bool armed = false, all = false;
  char name[64] = "";
  int stride = 1, seen = 0, launches = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;   // pre-created pool
  std::vector<int> call;                               // index of the sampled launch among the matching launches
  std::vector<const char*> names;                      // launcher name of each sample
};
KernelTimer g_kt;
}  // namespace

static int do_launch(const FdLaunch& L, hipStream_t stream) {
  void* argv[1] = {const_cast<char*>(L.arg.data())};
  int slot = -1;
  if (g_kt.armed && (g_kt.all || strcmp(L.name, g_kt.name) == 0)) {
    if (g_kt.seen % g_kt.stride == 0 && g_kt.call.size() < g_kt.ev.size()) {
      slot = (int)g_kt.call.size();
      g_kt.call.push_back(g_kt.seen);
      g_kt.names.push_back(L.name);
      (void)hipEventRecord(g_kt.ev[slot].first, stream);
    }
    ++g_kt.seen;
  }
  static const char* trace = FD_TUNE_GETENV("FDGAN_DEBUG_TRACE_LAUNCH");   // tuning build: name every launch and wait for it (a GPU
  if (trace) {                                                              // memory fault aborts the process: the last name is the culprit)
    fprintf(stderr, "[launch] %s grid=(%u,%u,%u) block=%u lds=%u\n", L.name, L.grid.x, L.grid.y, L.grid.z, L.block.x, L.shmem);
    fflush(stderr);
  }
  hipError_t e = hipLaunchKernel(L.fn, L.grid, L.block, argv, L.shmem, stream);
  if (trace && e == hipSuccess) e = hipStreamSynchronize(stream);
  if (slot >= 0) (void)hipEventRecord(g_kt.ev[slot].second, stream);
  if (e != hipSuccess) FD_FAIL(FD_ELAUNCH, "launch %s grid=(%u,%u,%u) block=%u lds=%u: %s", L.name, L.grid.x, L.grid.y,
                               L.grid.z, L.block.x, L.shmem, hipGetErrorString(e));
  return FD_OK;
}

extern "C" int fdgan_kernel_timer_arm(const char* name, int stride, int max_samples) {
  FD_REQUIRE(stride >= 1 && max_samples >= 1 && max_samples <= (1 << 16), "kernel_timer_arm: stride %d, max_samples %d", stride, max_samples);
  FD_REQUIRE(!g_kt.armed, "kernel_timer_arm: already armed (read it first)");
  g_kt.all = name == nullptr || strcmp(name, "*") == 0;
  strncpy(g_kt.name, g_kt.all ? "*" : name, sizeof(g_kt.name) - 1);
  g_kt.name[sizeof(g_kt.name) - 1] = 0;
  g_kt.stride = stride, g_kt.seen = 0;
  g_kt.call.clear(), g_kt.names.clear();
  while ((int)g_kt.ev.size() < max_samples) {
    hipEvent_t a = nullptr, b = nullptr;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) FD_FAIL(FD_ELAUNCH, "kernel_timer_arm: hipEventCreate failed");
    g_kt.ev.emplace_back(a, b);
  }
  while ((int)g_kt.ev.size() > max_samples) {
    hipEventDestroy(g_kt.ev.back().first), hipEventDestroy(g_kt.ev.back().second);
    g_kt.ev.pop_back();
  }
  g_kt.armed = true;
  return FD_OK;
}

extern "C" int fdgan_kernel_timer_read(int capacity, int* n_out, int* matching_launches_out, int* call_index, float* ms,
                                       char* names48) {
  FD_REQUIRE(g_kt.armed, "kernel_timer_read: not armed");
  FD_REQUIRE(n_out && capacity >= 0, "kernel_timer_read: bad arguments");
  g_kt.armed = false;
  const int n = (int)g_kt.call.size() < capacity ? (int)g_kt.call.size() : capacity;
  for (int i = 0; i < n; ++i) {
    hipError_t e = hipEventSynchronize(g_kt.ev[i].second);
    float t = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&t, g_kt.ev[i].first, g_kt.ev[i].second);
    if (e != hipSuccess) FD_FAIL(FD_ELAUNCH, "kernel_timer_read: %s", hipGetErrorString(e));
    if (ms) ms[i] = t;
    if (call_index) call_index[i] = g_kt.call[i];
    if (names48) {
      strncpy(names48 + 48 * i, g_kt.names[i], 47);
      names48[48 * i + 47] = 0;
    }
  }
  *n_out = n;
  if (matching_launches_out) *matching_launches_out = g_kt.seen;
  return FD_OK;
}

int fd_enqueue(const void* fn, const char* name, dim3 grid, dim3 block, unsigned shmem, const void* arg,
               size_t arg_bytes, hipStream_t stream) {
  if (grid.x == 0 || grid.y == 0 || grid.z == 0) FD_FAIL(FD_EINVAL, "%s: empty grid", name);
  FdLaunch L;
  L.fn = fn;
  L.name = name;
  L.grid = grid;
  L.block = block;
  L.shmem = shmem;
  L.arg.assign(static_cast<const char*>(arg), static_cast<const char*>(arg) + arg_bytes);
  if (g_recording != nullptr) {
    L.slot = g_recording->cur_slot;
    g_recording->launches.push_back(std::move(L));
    return FD_OK;
  }
  return do_launch(L, stream);
}

extern "C" FdPlan* fdgan_plan_create(void) { return new (std::nothrow) FdPlan(); }

extern "C" void fdgan_plan_destroy(FdPlan* p) {
  if (!p) return;
  if (g_recording == p) g_recording = nullptr;
  if (p->exec) hipGraphExecDestroy(p->exec);
  if (p->graph) hipGraphDestroy(p->graph);
  for (auto& pr : p->pairs) {
    hipEventDestroy(pr.first);
    hipEventDestroy(pr.second);
  }
  for (hipEvent_t e : p->wait_events) hipEventDestroy(e);
  delete p;
}

extern "C" int fdgan_plan_begin(FdPlan* p) {
  FD_REQUIRE(p, "plan_begin: NULL plan");
  if (g_recording != nullptr) FD_FAIL(FD_ESTATE, "plan_begin: another plan is recording on this thread");
  if (p->exec) FD_FAIL(FD_ESTATE, "plan_begin: plan already instantiated as a graph");
  p->recording = true;
  p->cur_slot = 0;
  g_recording = p;
  return FD_OK;
}

extern "C" int fdgan_plan_end(FdPlan* p) {
  FD_REQUIRE(p, "plan_end: NULL plan");
  if (g_recording != p) FD_FAIL(FD_ESTATE, "plan_end: this plan is not recording");
  p->recording = false;
  g_recording = nullptr;
  return FD_OK;
}

extern "C" int fdgan_plan_set_slot(FdPlan* p, int slot) {
  FD_REQUIRE(p && g_recording == p, "plan_set_slot: this plan is not recording");
  FD_REQUIRE(slot >= 0 && slot < 8, "plan_set_slot: slot %d outside [0, 8)", slot);
  p->cur_slot = slot;
  if (slot > p->max_slot) p->max_slot = slot;
  return FD_OK;
}

extern "C" int fdgan_plan_record_wait(FdPlan* p, int waiter_slot, int signaler_slot) {
  FD_REQUIRE(p && g_recording == p, "plan_record_wait: this plan is not recording");
  FD_REQUIRE(waiter_slot >= 0 && waiter_slot < 8 && signaler_slot >= 0 && signaler_slot < 8 && waiter_slot != signaler_slot,
             "plan_record_wait: slots %d <- %d", waiter_slot, signaler_slot);
  FdLaunch L;
  L.fn = nullptr;
  L.name = "stream_wait";
  L.shmem = 0;
  L.slot = waiter_slot;
  L.other = signaler_slot;
  if (waiter_slot > p->max_slot) p->max_slot = waiter_slot;
  if (signaler_slot > p->max_slot) p->max_slot = signaler_slot;
  p->launches.push_back(std::move(L));
  return FD_OK;
}

extern "C" int fdgan_plan_launch_multi(FdPlan* p, const FdStream* streams, int nstreams) {
  FD_REQUIRE(p && streams, "plan_launch_multi: NULL argument");
  if (p->recording) FD_FAIL(FD_ESTATE, "plan_launch_multi: plan is still recording");
  FD_REQUIRE(nstreams > p->max_slot, "plan_launch_multi: the plan uses stream slots 0..%d, %d streams given", p->max_slot, nstreams);
  size_t w = 0;
  for (const FdLaunch& L : p->launches) {
    if (L.fn != nullptr) {
      int rc = do_launch(L, static_cast<hipStream_t>(streams[L.slot]));
      if (rc != FD_OK) return rc;
      continue;
    }
    if (w == p->wait_events.size()) {
      hipEvent_t ev = nullptr;
      if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) FD_FAIL(FD_ELAUNCH, "plan_launch_multi: hipEventCreate failed");
      p->wait_events.push_back(ev);
    }
    hipEvent_t ev = p->wait_events[w++];
    hipError_t e = hipEventRecord(ev, static_cast<hipStream_t>(streams[L.other]));
    if (e == hipSuccess) e = hipStreamWaitEvent(static_cast<hipStream_t>(streams[L.slot]), ev, 0);
    if (e != hipSuccess) FD_FAIL(FD_ELAUNCH, "plan_launch_multi: stream dependency %d <- %d: %s", L.slot, L.other, hipGetErrorString(e));
  }
  return FD_OK;
}

extern "C" int64_t fdgan_plan_num_launches(const FdPlan* p) { return p ? (int64_t)p->launches.size() : -1; }

extern "C" const char* fdgan_plan_kernel_name(const FdPlan* p, int64_t k) {
  if (!p || k < 0 || k >= (int64_t)p->launches.size()) return nullptr;
  return p->launches[(size_t)k].name;
}

static int timed_launch(FdPlan* p, hipStream_t s) {
  // eager replay with a hipEvent pair around every marked launch (events are recorded on the
  // same stream the kernels run on; pairs accumulate until fdgan_plan_read_timing)
  size_t m = 0;
  for (size_t k = 0; k < p->launches.size(); ++k) {
    const bool marked = m < p->marked.size() && p->marked[m] == (int64_t)k;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (marked) {
      if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess)
        FD_FAIL(FD_ELAUNCH, "hipEventCreate failed");
      hipEventRecord(e0, s);
    }
    int rc = do_launch(p->launches[k], s);
    if (marked) {
      hipEventRecord(e1, s);
      p->pairs.push_back({e0, e1});
      ++m;
    }
    if (rc != FD_OK) return rc;
  }
  return FD_OK;
}

extern "C" int fdgan_plan_launch(FdPlan* p, FdStream stream) {
  FD_REQUIRE(p, "plan_launch: NULL plan");
  if (p->recording) FD_FAIL(FD_ESTATE, "plan_launch: plan is still recording");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (p->max_slot > 0) FD_FAIL(FD_ESTATE, "plan_launch: the plan records %d stream slots: use fdgan_plan_launch_multi", p->max_slot + 1);
  if (!p->marked.empty()) return timed_launch(p, s);
  if (p->exec) {
    hipError_t e = hipGraphLaunch(p->exec, s);
    if (e != hipSuccess) FD_FAIL(FD_ELAUNCH, "hipGraphLaunch: %s", hipGetErrorString(e));
    return FD_OK;
  }
  for (const FdLaunch& L : p->launches) {
    int rc = do_launch(L, s);
    if (rc != FD_OK) return rc;
  }
  return FD_OK;
}

extern "C" int fdgan_plan_instantiate_graph(FdPlan* p, FdStream stream) {
  FD_REQUIRE(p, "plan_instantiate_graph: NULL plan");
  if (p->recording) FD_FAIL(FD_ESTATE, "plan_instantiate_graph: plan is still recording");
  if (p->exec) return FD_OK;
  if (p->max_slot > 0) FD_FAIL(FD_ESTATE, "plan_instantiate_graph: multi-stream plans replay through fdgan_plan_launch_multi");
  // Capture never executes anything, so it runs on a private stream: the caller's stream may be
  // the legacy NULL stream, which cannot be captured.
  (void)stream;
  hipStream_t s = nullptr;
  hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  if (e != hipSuccess) FD_FAIL(FD_ELAUNCH, "hipStreamCreateWithFlags: %s", hipGetErrorString(e));
  e = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
  if (e != hipSuccess) {
    hipStreamDestroy(s);
    FD_FAIL(FD_ELAUNCH, "hipStreamBeginCapture: %s", hipGetErrorString(e));
  }
  int rc = FD_OK;
  for (const FdLaunch& L : p->launches) {
    rc = do_launch(L, s);
    if (rc != FD_OK) break;
  }
  hipGraph_t g = nullptr;
  e = hipStreamEndCapture(s, &g);
  hipStreamDestroy(s);
  if (rc != FD_OK) {
    if (g) hipGraphDestroy(g);
    return rc;
  }
  if (e != hipSuccess) FD_FAIL(FD_ELAUNCH, "hipStreamEndCapture: %s", hipGetErrorString(e));
  hipGraphExec_t ex = nullptr;
  e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
  if (e != hipSuccess) {
    hipGraphDestroy(g);
    FD_FAIL(FD_ELAUNCH, "hipGraphInstantiate: %s", hipGetErrorString(e));
  }
  p->graph = g;
  p->exec = ex;
  return FD_OK;
}

extern "C" int fdgan_plan_time_launches(FdPlan* p, const int64_t* idx, int64_t n) {
  FD_REQUIRE(p, "plan_time_launches: NULL plan");
  FD_REQUIRE(n == 0 || idx, "plan_time_launches: NULL index list");
  p->marked.clear();
  for (int64_t i = 0; i < n; ++i) {
    FD_REQUIRE(idx[i] >= 0 && idx[i] < (int64_t)p->launches.size(), "plan_time_launches: index %lld out of range",
               (long long)idx[i]);
    FD_REQUIRE(i == 0 || idx[i] > idx[i - 1], "plan_time_launches: indices must be strictly increasing");
    p->marked.push_back(idx[i]);
  }
  return FD_OK;
}

extern "C" int fdgan_plan_read_timing(FdPlan* p, double* total_ms, int64_t* launches) {
  FD_REQUIRE(p && total_ms && launches, "plan_read_timing: NULL argument");
  double tot = 0.0;
  int64_t cnt = 0;
  int rc = FD_OK;
  for (auto& pr : p->pairs) {
    float ms = 0.f;
    hipError_t e = hipEventSynchronize(pr.second);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, pr.first, pr.second);
    if (e != hipSuccess) {
      fd_set_error("plan_read_timing: %s", hipGetErrorString(e));
      rc = FD_ELAUNCH;
    } else {
      tot += ms;
      ++cnt;
    }
    hipEventDestroy(pr.first);
    hipEventDestroy(pr.second);
  }
  p->pairs.clear();
  *total_ms = tot;
  *launches = cnt;
  return rc;
}

extern "C" int fdgan_plan_profile(FdPlan* p, FdStream stream, float* ms_out, int64_t n) {
  FD_REQUIRE(p && ms_out, "plan_profile: NULL argument");
  if (p->recording) FD_FAIL(FD_ESTATE, "plan_profile: plan is still recording");
  if (p->max_slot > 0) FD_FAIL(FD_ESTATE, "plan_profile: single-stream plans only");
  FD_REQUIRE(n == (int64_t)p->launches.size(), "plan_profile: ms_out holds %lld entries, plan has %zu launches",
             (long long)n, p->launches.size());
  hipStream_t s = static_cast<hipStream_t>(stream);
  std::vector<hipEvent_t> ev(p->launches.size() + 1, nullptr);
  int rc = FD_OK;
  for (auto& e : ev)
    if (hipEventCreate(&e) != hipSuccess) rc = FD_ELAUNCH;
  if (rc == FD_OK) {
    hipEventRecord(ev[0], s);
    for (size_t k = 0; k < p->launches.size() && rc == FD_OK; ++k) {
      rc = do_launch(p->launches[k], s);
      hipEventRecord(ev[k + 1], s);
    }
    hipStreamSynchronize(s);
    for (size_t k = 0; k < p->launches.size() && rc == FD_OK; ++k)
      if (hipEventElapsedTime(&ms_out[k], ev[k], ev[k + 1]) != hipSuccess) {
        fd_set_error("plan_profile: hipEventElapsedTime failed at launch %zu", k);
        rc = FD_ELAUNCH;
      }
  } else {
    fd_set_error("plan_profile: hipEventCreate failed");
  }
  for (auto& e : ev)
    if (e) hipEventDestroy(e);
  return rc;
}
