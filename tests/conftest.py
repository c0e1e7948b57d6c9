import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "fd-gan_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)
GOLDEN = os.path.join(ROOT, "tests", "golden")

# FDGAN_TEST_GUARD_ALLOC=<libguard_alloc.so> (tools/dbg/guard_alloc.cpp): every device tensor gets its own virtual-memory reservation
# that ENDS at an unmapped guard range -- an access 16 bytes past the end of any buffer faults at the offending launch (run with
# AMD_SERIALIZE_KERNEL=3 and a tuning build's FDGAN_DEBUG_TRACE_LAUNCH to have it named).  Must be installed before the first allocation.
if os.environ.get("FDGAN_TEST_GUARD_ALLOC"):
    import torch
    _guard = torch.cuda.memory.CUDAPluggableAllocator(os.environ["FDGAN_TEST_GUARD_ALLOC"], "guard_malloc", "guard_free")
    torch.cuda.memory.change_current_allocator(_guard)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


# FDGAN_TEST_POISON=1: every `torch.empty` / `empty_like` on the GPU comes back filled with a huge FINITE value (1e30; 3e4 in fp16; all-ones
# bytes for integers) instead of whatever the caching allocator's block held before: a kernel that consumes a workspace slot, a partial sum
# or a temporary nobody wrote shows up as a 1e30-sized error in EVERY run, not as a NaN in one run of six (finite on purpose: padded channels
# are legitimately multiplied by zero filter entries, and 0 x NaN would flag those).
if os.environ.get("FDGAN_TEST_POISON"):
    import torch as _torch
    _empty, _empty_like = _torch.empty, _torch.empty_like

    def _poison(t):
        if t.is_cuda and t.numel():
            if t.dtype == _torch.float16:
                t.fill_(3.0e4)
            elif t.dtype.is_floating_point:
                t.fill_(1.0e30)
            elif t.dtype == _torch.bool:
                t.fill_(True)
            elif not t.dtype.is_complex:
                t.view(_torch.uint8).fill_(255) if t.is_contiguous() else None
        return t
    _torch.empty = lambda *a, **k: _poison(_empty(*a, **k))
    _torch.empty_like = lambda *a, **k: _poison(_empty_like(*a, **k))

# FDGAN_TEST_MEMTRACE=<dir> (fault forensics, tools/dbg/memtrace_lookup.py): torch's allocator history is recorded with Python
# stacks and a daemon thread writes a compact snapshot -- every segment with its blocks, the last few thousand alloc / free /
# segment events, the running test -- once a second, so that after a "Memory access fault by GPU ... on address X" (the process
# is aborted from a runtime thread: no Python code runs any more) the buffer that ended or began at X can be named.
_MEMTRACE = os.environ.get("FDGAN_TEST_MEMTRACE")
_CURRENT_TEST = ["<none>"]
if _MEMTRACE:
    import json
    import threading
    import time
    import torch
    os.makedirs(_MEMTRACE, exist_ok=True)
    torch.cuda.memory._record_memory_history(max_entries=60000, context="alloc", stacks="python")

    def _frames(ev):
        out = []
        for fr in ev.get("frames", [])[:40]:
            fn = fr.get("filename", "")
            if "site-packages" in fn or "dist-packages" in fn or fn.startswith("/usr/lib") or fn.startswith("<"):
                continue
            out.append("%s:%s:%s" % (os.path.basename(fn), fr.get("line"), fr.get("name")))
            if len(out) == 6:
                break
        return out

    def _memtrace_loop():
        i = 0
        while True:
            time.sleep(1.0)
            try:
                snap = torch.cuda.memory._snapshot()
                segs = [[sg["address"], sg["total_size"], sg.get("stream", 0),
                         [[b["address"] if "address" in b else None, b["size"], b["state"]] for b in sg["blocks"]]] for sg in snap["segments"]]
                tr = snap["device_traces"][0] if snap.get("device_traces") else []
                evs = [[e["action"], e.get("addr"), e.get("size"), e.get("stream"), _frames(e)] for e in tr[-6000:]]
                tmp = os.path.join(_MEMTRACE, "snap_%d.tmp" % (i & 1))
                with open(tmp, "w") as f:
                    json.dump({"test": _CURRENT_TEST[0], "time": time.time(), "segments": segs, "events": evs}, f)
                os.replace(tmp, os.path.join(_MEMTRACE, "snap_%d.json" % (i & 1)))
                i += 1
            except Exception as e:      # never disturb the tests
                with open(os.path.join(_MEMTRACE, "error.txt"), "a") as f:
                    f.write(repr(e) + "\n")
    threading.Thread(target=_memtrace_loop, daemon=True).start()


@pytest.fixture(autouse=True)
def _current_test_name(request):
    _CURRENT_TEST[0] = request.node.nodeid
    yield


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def manifest():
    import json
    with open(os.path.join(GOLDEN, "MANIFEST.json")) as f:
        return json.load(f)


@pytest.fixture(autouse=True)
def _gpu_memory_log(request):
    """FDGAN_TEST_MEMLOG=<file>: after every GPU test, one line with the device's free / torch-reserved / torch-allocated GiB."""
    yield
    path = os.environ.get("FDGAN_TEST_MEMLOG")
    if not path or request.node.get_closest_marker("gpu") is None:
        return
    import torch
    if torch.cuda.is_available():
        free, total = torch.cuda.mem_get_info()
        with open(path, "a") as f:
            st = torch.cuda.memory_stats()
            f.write("%-90s free %7.2f  reserved %7.2f  allocated %7.2f GiB  retries %d ooms %d segments %d\n" % (
                request.node.nodeid[-90:], free / 2**30, torch.cuda.memory_reserved() / 2**30, torch.cuda.memory_allocated() / 2**30,
                st.get("num_alloc_retries", -1), st.get("num_ooms", -1), st.get("segment.all.current", -1)))


@pytest.fixture(autouse=True)
def _gpu_test_hygiene(request):
    """Every GPU test starts with the previous tests' garbage collected (plans, tapes, 0.6 GB of workspaces per reverse walk: reference
    cycles, otherwise finalised at a random point INSIDE a later test) and ends with an idle device.  FDGAN_TEST_HYGIENE=gc / sync /
    none: one half only / neither (to study the fault this fixture keeps away, see DESIGN.md status round 5 #10)."""
    mode = os.environ.get("FDGAN_TEST_HYGIENE", "both")
    gpu = request.node.get_closest_marker("gpu") is not None
    if gpu and mode in ("both", "gc"):
        import gc
        gc.collect()
    yield
    if gpu and mode in ("both", "sync"):
        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize()
