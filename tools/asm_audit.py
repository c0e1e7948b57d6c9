"""Which kernels wait for a load right behind it?  (experiment aid, no GPU needed)
Compiles every csrc/*.hip to gfx950 assembly and counts, per kernel, the global loads that are followed by `s_waitcnt vmcnt(0)`
within 14 instructions with no other load or MFMA in between: the signature of a load inside a branch (`ok ? *p : 0`,
`if (more) prefetch()`), which hipcc waits for at the join instead of at its first use (Status round 4 #7).
Prologue loads (per-channel coefficients) show up too; the interesting ones sit in loops.     python tools/asm_audit.py [min_count]"""
import glob, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W = 14
out = tempfile.mkdtemp(prefix="fdasm")
procs = []
for f in sorted(glob.glob(os.path.join(ROOT, "fd-gan_amd", "csrc", "*.hip"))):
    s = os.path.join(out, os.path.basename(f)[:-4] + ".s")
    procs.append((s, subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "-mllvm",
                                       "-pragma-unroll-threshold=200000", "-S", "--cuda-device-only", "-I" + os.path.join(ROOT, "include"), "-o", s, f],
                                      stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)))
minc = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for s, p in procs:
    p.wait()
    if not os.path.exists(s):
        continue
    lines = open(s).read().split("\n")
    cur, res = None, {}
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            cur = m.group(1); res[cur] = [0, 0, 0]; continue
        if cur is None:
            continue
        if "s_endpgm" in l:
            cur = None; continue
        if "v_mfma" in l:
            res[cur][2] += 1
        if re.search(r"\bglobal_load_", l):
            res[cur][0] += 1
            k, n = i + 1, 0
            while k < len(lines) and n < W:
                t = lines[k].strip(); k += 1
                if not t or t[0] in ";.":
                    continue
                if re.search(r"\bglobal_load_", t) or "v_mfma" in t:
                    break
                if "s_waitcnt vmcnt(0)" in t:
                    res[cur][1] += 1; break
                n += 1
    for k, (nl, nw, nm) in res.items():
        if nw >= minc:
            print("%-22s %-84s loads %4d  load->vmcnt(0) %3d  mfma %d" % (os.path.basename(s), k[:84], nl, nw, nm))
