"""Instruction mix of a kernel's loops (experiment aid, no GPU needed).
    python tools/isa_mix.py <file.s | csrc/file.hip> <kernel-name substring> [-D...]
For every loop of the kernel (a backward branch to a label), the instruction count by class -- MFMA, packed / conversion / other
VALU, LDS reads / writes, global loads / stores, waits, barriers, SALU -- and the VALU : MFMA ratio.  The steady-state loop of a
hand-scheduled kernel is its largest loop."""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def compile_s(src, extra):
    out = os.path.join(tempfile.mkdtemp(prefix="isamix"), "k.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "-Wno-unused-result",
                           "-mllvm", "-pragma-unroll-threshold=200000", "-S", "--cuda-device-only", "-I" + os.path.join(ROOT, "include"),
                           "-o", out, src] + extra, stderr=subprocess.DEVNULL)
    return out


def klass(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_pk_"): return "valu_pk"
    if op.startswith("v_cvt"): return "valu_cvt"
    if op.startswith("v_cmp") or op.startswith("v_cndmask"): return "valu_sel"
    if op.startswith(("v_readfirstlane", "v_readlane", "v_writelane")): return "valu_lane"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_read") or op.startswith("ds_load"): return "lds_rd"
    if op.startswith("ds_"): return "lds_wr"
    if op.startswith(("global_load", "buffer_load", "scratch_load", "flat_load")): return "g_ld"
    if op.startswith(("global_store", "buffer_store", "scratch_store", "flat_store", "global_atomic")): return "g_st"
    if op == "s_waitcnt": return "wait"
    if op == "s_barrier": return "barrier"
    if op == "s_nop": return "nop"
    if op.startswith("s_"): return "salu"
    return "other"


def main():
    src, name = sys.argv[1], sys.argv[2]
    path = src if src.endswith(".s") else compile_s(src, sys.argv[3:])
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l) and name in l)
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start:end]
    labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
    loops = []
    for i, l in enumerate(body):
        m = re.search(r"\bs_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((labels[m.group(1)], i))
    tot = collections.Counter()
    for l in body:
        t = l.strip().split()
        if t and not t[0].startswith((";", ".")) and not t[0].endswith(":"):
            tot[klass(t[0])] += 1
    print("kernel %s: %d instructions  %s" % (lines[start].split(":")[0], sum(tot.values()), dict(tot)))
    for a, b in sorted(loops, key=lambda ab: ab[0] - ab[1])[:4]:
        c, ops = collections.Counter(), collections.Counter()
        for l in body[a:b + 1]:
            t = l.strip().split()
            if t and not t[0].startswith((";", ".")) and not t[0].endswith(":"):
                c[klass(t[0])] += 1
                ops[t[0]] += 1
        valu = sum(v for k, v in c.items() if k.startswith("valu"))
        print("loop lines %d-%d: %d instr, VALU %d, MFMA %d (%.1f VALU/MFMA)  %s" % (a, b, sum(c.values()), valu, c["mfma"], valu / max(c["mfma"], 1), dict(c)))
        print("   top ops:", ", ".join("%s x%d" % kv for kv in ops.most_common(28)))
        waits = [l.strip() for l in body[a:b + 1] if "s_waitcnt" in l]
        print("   waits:", collections.Counter(waits).most_common(12))


if __name__ == "__main__":
    main()
