"""Out-of-bounds regression (round 6).  Round 5's intermittent "Memory access fault by GPU" was `conv_wgrad_tr`'s masked units reading
their (unclamped) channel piece at the clamped pixel 0: up to 32 bytes past the END of a 512-byte activation of the legacy U-Nets'
1 x 1 stage -- harmless unless torch's caching allocator had put that tensor at the very end of a segment, so no functional test can
see it.  This one can: the offending tests run in a child process whose every device tensor ENDS at an unmapped page
(tools/dbg/guard_alloc.cpp as torch's pluggable allocator, GUARD_ALLOC_END=1 GUARD_ALLOC_LEAK=1); an access >= 16 bytes past the end of
any buffer aborts the child."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


def _guard_lib(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc on this box: the guard allocator is built where it runs")
    so = str(tmp_path_factory.mktemp("guard") / "libguard_alloc.so")
    r = subprocess.run([HIPCC, "-O2", "-w", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tools", "dbg", "guard_alloc.cpp")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return so


@pytest.mark.gpu
def test_guard_allocator_is_sound_and_catches_an_overrun(tmp_path_factory):
    """The tool itself: torch fills / copies / kernels are right under it, and a deliberate read 16 bytes past a tensor's end aborts."""
    so = _guard_lib(tmp_path_factory)
    env = dict(os.environ, GUARD_ALLOC_END="1", GUARD_ALLOC_LEAK="1")
    ok = ("import torch\n"
          "a = torch.cuda.memory.CUDAPluggableAllocator(%r, 'guard_malloc', 'guard_free')\n"
          "torch.cuda.memory.change_current_allocator(a)\n"
          "x = torch.arange(3000, dtype=torch.float32).cuda()\n"
          "assert bool(((x * 2).cpu() == torch.arange(3000) * 2.0).all())\n" % so)
    r = subprocess.run([sys.executable, "-c", ok], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    # a library launch told to zero 64 bytes more than the tensor has (torch's own ops refuse an out-of-bounds view)
    bad = ok + ("import sys\nsys.path[:0] = [%r, %r]\n"
                "from fdgan_hip import lib as L\n"
                "L.check(L.load().fdgan_fill_zero(x.data_ptr(), 12000 + 64, 1, 0, None), 'fill_zero')\n"
                "torch.cuda.synchronize()\n" % (ROOT, os.path.join(ROOT, "fd-gan_amd")))
    r = subprocess.run([sys.executable, "-c", bad], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "Memory access fault" in (r.stderr + r.stdout), (r.returncode, r.stderr[-1500:])


@pytest.mark.gpu
def test_no_kernel_reads_past_the_end_of_a_buffer_legacy_backward(tmp_path_factory):
    so = _guard_lib(tmp_path_factory)
    env = dict(os.environ, FDGAN_TEST_GUARD_ALLOC=so, GUARD_ALLOC_END="1", GUARD_ALLOC_LEAK="1", FDGAN_TEST_HYGIENE="none")
    env.pop("FDGAN_TEST_MEMTRACE", None)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_hip_models.py"), "-m", "gpu", "-q", "-x",
                        "-p", "no:cacheprovider", "-k", "legacy_unets_backward or legacy_dehaze_backward or dehaze22_d_backward"],
                       env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    tail = (r.stdout[-1500:] + "\n" + r.stderr[-1500:])
    assert r.returncode == 0 and "passed" in r.stdout, tail
