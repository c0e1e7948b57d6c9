"""CPU tests: the C-ABI library loads and exports every symbol include/fdgan_hip.h
declares, host-only entry points behave, and the nn.Module surface is key-compatible
with the reference (via the oracle) and fails loudly without a GPU."""
import ctypes as C
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from fdgan_hip import lib as L
    hdr = open(os.path.join(ROOT, "include", "fdgan_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(fdgan_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(L.SIGNATURES), declared ^ set(L.SIGNATURES)
    lib = L.load()
    raw = C.CDLL(L.LIB_PATH)
    for name in declared:
        assert getattr(raw, name) is not None
    assert lib.fdgan_version() == L.ABI_VERSION


def test_stale_library_is_refused(monkeypatch, tmp_path):
    """the library carries the hash of the sources it was compiled from, and the binding refuses one whose
    sources have changed since (the library is git-ignored and rebuilt by mtime; FDGAN_ABI_VERSION is bumped by hand)."""
    import shutil
    from fdgan_hip import buildid
    from fdgan_hip import lib as L
    lib = L.load()
    built = lib.fdgan_build_id().decode()
    assert re.fullmatch(r"[0-9a-f]{16}(:.*)?", built), built
    assert built.split(":")[0] == buildid.source_build_id()
    # the hash covers every translation unit, every header of csrc/ and the public header, by name and content
    files = [os.path.basename(f) for f in buildid.source_files()]
    assert "conv1x1_bwd.hip" in files and "conv_igemm.h" in files and files[-1] == "fdgan_hip.h"
    csrc = tmp_path / "csrc"
    shutil.copytree(buildid.CSRC, csrc)
    hdr = tmp_path / "fdgan_hip.h"
    shutil.copy(buildid.HEADER, hdr)
    assert buildid.source_build_id(str(csrc), str(hdr)) == built.split(":")[0]
    with open(csrc / "common.h", "a") as f:
        f.write("// an edit after the build\n")
    changed = buildid.source_build_id(str(csrc), str(hdr))
    assert changed != built.split(":")[0]
    os.rename(csrc / "pool.hip", csrc / "pool2.hip")
    assert buildid.source_build_id(str(csrc), str(hdr)) not in (changed, built.split(":")[0])
    # a tree that differs from the loaded library's: load() fails loudly, nothing is bound
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(buildid, "source_build_id", lambda *a: changed)
    with pytest.raises(L.FdganLibraryError, match="stale"):
        L.load()
    assert L._lib is None
    monkeypatch.setenv("FDGAN_ALLOW_STALE_LIB", "1")      # tools/ only: A/B against a variant built from another commit
    assert L.load() is not None
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(buildid, "source_build_id", lambda *a: None)   # an installed library without its sources: nothing to compare
    monkeypatch.delenv("FDGAN_ALLOW_STALE_LIB")
    assert L.load() is not None


def test_host_only_entry_points():
    from fdgan_hip import engine as E
    from fdgan_hip import lib as L
    lib = L.load()
    assert lib.fdgan_packed_weight_bytes(32, 128, 3) == 4 * 9 * 2 * 1024          # chunks*taps*tiles KiB
    assert lib.fdgan_packed_weight_bytes(3, 16, 3) == 1 * 9 * 1 * 1024
    assert lib.fdgan_packed_weight_bytes(0, 16, 3) == 0
    # fwd_info is a dry run (no HIP call): grid of the headline shape B=16 @256^2, 128->32 3x3
    x = torch.empty((16, 256, 256, 128), dtype=torch.float16, device="meta")

    def fd(n, h, w, c, pitch):
        t = L.FdTensor()
        t.ptr, t.n, t.h, t.w, t.c = 4096, n, h, w, c
        t.stride[0], t.stride[1], t.stride[2], t.stride[3] = h * w * pitch, w * pitch, pitch, 1
        t.dtype = L.FD_F16
        return t
    info = E.conv_info(fd(16, 256, 256, 128, 128), fd(16, 256, 256, 32, 256), 32, E.conv_desc(3, 1, 1))
    # persistent-filter kernel: one 8-wave workgroup per CU, one statistics row per workgroup
    assert (info.grid_x, info.grid_y, info.stats_rows, info.stats_cpad) == (256, 1, 256, 32)
    assert info.lds_bytes <= 160 * 1024
    assert lib.fdgan_conv_weight_layout(128, 256, 1, 1) == L.WLAYOUT_X64
    # CU budget (ABI v16): the persistent kernels size their grid for the CUs of a masked stream; thread-local, restored on exit
    with E.cu_budget(192):
        info192 = E.conv_info(fd(16, 256, 256, 128, 128), fd(16, 256, 256, 32, 256), 32, E.conv_desc(3, 1, 1))
        assert (info192.grid_x, info192.stats_rows) == (192, 192)
        with E.cu_budget(0):
            assert E.conv_info(fd(16, 256, 256, 128, 128), fd(16, 256, 256, 32, 256), 32, E.conv_desc(3, 1, 1)).grid_x == 256
        assert lib.fdgan_set_cu_budget(192) == 192
    assert lib.fdgan_set_cu_budget(0) == 0
    assert E.conv_info(fd(16, 256, 256, 128, 128), fd(16, 256, 256, 32, 256), 32, E.conv_desc(3, 1, 1)).grid_x == 256
    info = E.conv_info(fd(16, 256, 256, 256, 256), fd(16, 128, 128, 128, 160), 128,
                       E.conv_desc(1, w_layout=L.WLAYOUT_X64),
                       E.make_prologue(pool=True))
    assert (info.grid_x, info.grid_y) == (512, 1)                                   # persistent x-stream kernel
    # argument validation crosses the ABI as an error code + message, never an abort
    bad = fd(1, 8, 8, 16, 12)
    rc = lib.fdgan_conv2d_fwd_info(C.byref(bad), C.byref(fd(1, 8, 8, 16, 16)), 16, C.byref(E.conv_desc(3, 1, 1)), None,
                                   C.byref(L.FdConvInfo()))
    assert rc == L.FD_EINVAL and b"multiples of 8" in lib.fdgan_last_error()
    rc = lib.fdgan_conv2d_fwd_info(C.byref(fd(1, 8, 8, 16, 16)), C.byref(fd(1, 8, 8, 16, 16)), 16,
                                   C.byref(E.conv_desc(5, 1, 2)), None, C.byref(L.FdConvInfo()))
    assert rc == L.FD_EINVAL
    # plan API misuse
    p = lib.fdgan_plan_create()
    assert lib.fdgan_plan_end(p) == L.FD_ESTATE
    assert lib.fdgan_plan_begin(p) == L.FD_OK and lib.fdgan_plan_begin(p) == L.FD_ESTATE
    assert lib.fdgan_plan_launch(p, None) == L.FD_ESTATE
    assert lib.fdgan_plan_end(p) == L.FD_OK and lib.fdgan_plan_num_launches(p) == 0
    lib.fdgan_plan_destroy(p)


def test_module_surface_is_reference_compatible():
    import models.dehaze1113 as net
    from oracle import dehaze1113_ref as ref
    g, og = net.FDGAN(), ref.FDGAN()
    assert list(g.state_dict().keys()) == list(og.state_dict().keys())
    for (k, a), (_, b) in zip(g.state_dict().items(), og.state_dict().items()):
        assert a.shape == b.shape and a.dtype == b.dtype, k
    assert g.training and sum(p.numel() for p in g.parameters()) == 13980691
    d, od = net.D(9, 36), ref.D(9, 36)
    assert list(d.state_dict().keys()) == list(od.state_dict().keys())
    assert sum(p.numel() for p in d.parameters()) == 790416
    import models.dehaze22 as net22
    from oracle import dehaze22_ref as ref22
    assert list(net22.D(6, 64).state_dict().keys()) == list(ref22.D(6, 64).state_dict().keys())
    b = net.BottleneckBlockdy(64, 32)
    assert set(b.state_dict()) == set(ref.BottleneckBlockdy(64, 32).state_dict())
    assert tuple(net.TransitionBlockdy(96, 16).conv1.weight.shape) == (96, 16, 1, 1)


def test_no_cpu_fallback():
    import models.dehaze1113 as net
    g = net.FDGAN()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        g(torch.zeros(1, 3, 64, 64))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net.D(9, 36)(torch.zeros(1, 9, 64, 64))
    with pytest.raises(RuntimeError):
        g.dense_block1.denselayer1(torch.zeros(1, 64, 8, 8))       # containers never compute


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under fd-gan_amd/ may reference it."""
    pkg = os.path.join(ROOT, "fd-gan_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith(".py"):
                src = open(os.path.join(dp, fn)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), os.path.join(dp, fn)


def test_release_library_carries_no_tuning_switches():
    """The FDGAN_DEBUG_* switches (kernel selection, phase skipping -- some make a kernel return wrong results) exist only
    in -DFDGAN_TUNING builds (csrc/common.h: FD_TUNE_GETENV): the library the tests and the benchmark load must not contain
    their names, and must not look at the environment for them."""
    from fdgan_hip import lib as L
    blob = open(L.LIB_PATH, "rb").read()
    assert b"FDGAN_DEBUG_" not in blob and b"FDGAN_TUNING" not in blob
    flags = os.path.join(ROOT, "fd-gan_amd", "build", ".flags")
    if os.path.exists(flags):
        assert "-DFDGAN_TUNING" not in open(flags).read()
    for dp, _, fns in os.walk(os.path.join(ROOT, "fd-gan_amd", "csrc")):
        for fn in fns:
            src = open(os.path.join(dp, fn)).read()
            for m in re.finditer(r"(?<![A-Z_])getenv\s*\(", src):
                line = src[src.rfind("\n", 0, m.start()) + 1:src.find("\n", m.start())]
                assert "#define FD_TUNE_GETENV" in line, "%s: getenv outside FD_TUNE_GETENV: %s" % (fn, line.strip())


def test_no_kernel_spills_outside_the_allow_list():
    """spills have silently come and gone in three rounds (a scratch reload is a `vmcnt(0)` in a streaming
    kernel).  Every csrc/*.hip is compiled to gfx950 assembly with the release flags; a kernel may contain `scratch_`
    instructions only if tools/scratch_allow.json names it, and then at most the count recorded there."""
    import json
    import shutil
    import sys
    if shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import scratch_audit
    allow = json.load(open(os.path.join(ROOT, "tools", "scratch_allow.json")))
    res = scratch_audit.audit()
    bad = []
    seen = set()
    for f, per in res.items():
        for k, v in per.items():
            n = max(v["scratch"], 1 if (v["private"] or v["spill"]) else 0)
            if n == 0:
                continue
            seen.add(k)
            if k not in allow["kernels"]:
                bad.append("%s: %s has %d scratch instructions (%d B private) and is not on the allow-list" % (f, k, v["scratch"], v["private"]))
            elif v["scratch"] > allow["kernels"][k]["max_scratch"]:
                bad.append("%s: %s has %d scratch instructions, allow-list says <= %d" % (f, k, v["scratch"], allow["kernels"][k]["max_scratch"]))
    assert not bad, "\n".join(bad)
    # a stale allow-list entry (the kernel is clean now or gone) is an error too: the list may only shrink
    stale = sorted(set(allow["kernels"]) - seen)
    assert not stale, "allow-list entries that no longer spill (remove them): %s" % stale
