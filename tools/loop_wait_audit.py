"""Which kernels drain their prefetch inside the main loop?  (no GPU needed)
Compiles csrc/*.hip to gfx950 assembly and reports, per kernel, the loops (LLVM's `in Loop: Header=` block annotations) whose body
holds global loads AND an `s_waitcnt vmcnt(0)` (or a `__syncthreads()`-style vmcnt(0) in front of s_barrier): there the loads just
issued are waited for inside the same iteration -- no overlap with the iteration's compute.  tools/asm_audit.py looks at single loads;
this one found conv_wgrad_tr's row loop (round 5), whose four loads sat in an `if` and were drained right behind it.
    python tools/loop_wait_audit.py [file.hip ...]"""
import glob, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
files = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "fd-gan_amd", "csrc", "*.hip")))
out = tempfile.mkdtemp(prefix="fdloop")
procs = []
for f in files:
    s = os.path.join(out, os.path.basename(f)[:-4] + ".s")
    procs.append((s, subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "-mllvm",
                                       "-pragma-unroll-threshold=200000", "-S", "--cuda-device-only", "-I" + os.path.join(ROOT, "include"), "-o", s, f],
                                      stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)))
for s, p in procs:
    p.wait()
    if not os.path.exists(s):
        continue
    kern, loop = None, None
    stats = {}
    for l in open(s):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            kern, loop = m.group(1), None
            continue
        if kern is None:
            continue
        m = re.match(r"^\.LBB\d+_\d+:\s*;\s*(.*)", l)
        if m:
            c = m.group(1)
            mm = re.search(r"Header=(BB\d+_\d+)", c)
            if "Loop Header" in c:
                loop = re.match(r"^\.L(BB\d+_\d+)", l).group(1)
            elif mm:
                loop = mm.group(1)
            else:
                loop = None
            continue
        if re.match(r"^\.LBB\d+_\d+:", l):
            loop = None
            continue
        if loop is None:
            continue
        t = l.strip()
        st = stats.setdefault((kern, loop), [0, 0, 0, 0, 0, 0])      # loads, mfma, vmcnt(0), scratch reloads, EARLY drains, pending loads since the last mfma
        if re.match(r"global_load_|buffer_load_", t):
            st[0] += 1
            st[5] += 1
        elif t.startswith("v_mfma"):
            st[1] += 1
            st[5] = 0
        elif t.startswith("s_waitcnt") and "vmcnt(0)" in t:
            st[2] += 1
            if st[5] > 0:      # loads issued since the last MFMA are drained before any compute covers them
                st[4] += 1
            st[5] = 0
        elif t.startswith("scratch_load"):
            st[3] += 1
    for (k, lp), (nl, nm, nw, ns, ne, _) in sorted(stats.items()):
        if nl > 0 and (ne > 0 or ns > 0) and nm > 0:
            print("%-22s %-86s loop %-9s loads %3d  mfma %4d  vmcnt(0) %2d (right behind loads: %d)  scratch reloads %2d" % (os.path.basename(s), k[:86], lp, nl, nm, nw, ne, ns))
