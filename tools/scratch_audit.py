"""Which kernels spill?  (no GPU needed)
Compiles every csrc/*.hip to gfx950 assembly with the release flags and reports, per kernel, the number of `scratch_`
instructions and the kernel descriptor's `.private_segment_fixed_size` / `.vgpr_spill_count`.  A scratch reload is a
`s_waitcnt vmcnt(0)` in a streaming kernel, so every hot kernel has to stay at zero;
tests/test_capi_cpu.py::test_no_kernel_spills_outside_the_allow_list runs this against tools/scratch_allow.json.

    python tools/scratch_audit.py            # table of every kernel that touches scratch
    python tools/scratch_audit.py --json     # {file: {kernel: {"scratch": n, "private": bytes, "spill": n}}}
"""
import glob, json, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "-Wno-unused-result", "-mllvm",
         "-pragma-unroll-threshold=200000"]


def demangle(names):
    try:
        r = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"], input="\n".join(names), capture_output=True, text=True)
        out = r.stdout.split("\n")
        return dict(zip(names, out))
    except OSError:
        return {n: n for n in names}


def audit(files=None, jobs=8):
    out = tempfile.mkdtemp(prefix="fdscr")
    files = files or sorted(glob.glob(os.path.join(ROOT, "fd-gan_amd", "csrc", "*.hip")))
    res = {}
    pending = list(files)
    running = []
    done = []
    while pending or running:
        while pending and len(running) < jobs:
            f = pending.pop(0)
            s = os.path.join(out, os.path.basename(f)[:-4] + ".s")
            running.append((f, s, subprocess.Popen([HIPCC] + FLAGS + ["-S", "--cuda-device-only", "-I" + os.path.join(ROOT, "include"), "-o", s, f],
                                                   stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)))
        f, s, p = running.pop(0)
        err = p.communicate()[1]
        if p.returncode != 0:
            raise RuntimeError("hipcc -S failed on %s:\n%s" % (f, err[-3000:]))
        done.append((f, s))
    for f, s in done:
        lines = open(s).read().split("\n")
        cur, per = None, {}
        for l in lines:
            m = re.match(r"^(_Z\w+):", l)
            if m:
                cur = m.group(1)
                per[cur] = {"scratch": 0, "private": 0, "spill": 0}
                continue
            if cur is not None:
                if "s_endpgm" in l:
                    # a kernel may have several exits; keep counting until the next label of a kernel
                    pass
                t = l.strip()
                if t.startswith("scratch_"):
                    per[cur]["scratch"] += 1
        # descriptors: .amdhsa_kernel NAME ... / metadata
        txt = "\n".join(lines)
        for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, re.S):
            k, body = m.group(1), m.group(2)
            mm = re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", body)
            if k in per and mm:
                per[k]["private"] = int(mm.group(1))
        for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.vgpr_spill_count:\s+(\d+)", txt):
            if m.group(1) in per:
                per[m.group(1)]["spill"] = int(m.group(2))
        res[os.path.basename(f)] = per
    return res


def main():
    res = audit()
    if "--json" in sys.argv:
        print(json.dumps(res, indent=1, sort_keys=True))
        return
    names = [k for per in res.values() for k in per]
    dm = demangle(names)
    total = 0
    for f, per in sorted(res.items()):
        for k, v in sorted(per.items()):
            if v["scratch"] or v["private"] or v["spill"]:
                total += v["scratch"]
                print("%-22s scratch %4d  private %5d B  vgpr_spill %3d  %s" % (f, v["scratch"], v["private"], v["spill"], dm.get(k, k)[:110]))
    print("total scratch instructions: %d in %d kernels" % (total, sum(1 for per in res.values() for v in per.values() if v["scratch"])))


if __name__ == "__main__":
    main()
