"""Post-mortem of a GPU memory fault with the snapshots tests/conftest.py writes under FDGAN_TEST_MEMTRACE (debugging aid):
    python tools/dbg/memtrace_lookup.py 0x7c3c3ac00000 gpurun_out/memtrace/snap_0.json [snap_1.json]
Prints the segment(s) that contain, end at or begin at the address, their blocks around it, and every recorded allocator event
(with the Python frames of the allocation) on addresses within 64 MiB of it."""
import json, sys
addr = int(sys.argv[1], 16)
for path in sys.argv[2:]:
    d = json.load(open(path))
    print("== %s  (running test: %s)" % (path, d["test"]))
    for a, size, stream, blocks in sorted(d["segments"]):
        if a - (64 << 20) <= addr <= a + size + (64 << 20):
            rel = "CONTAINS" if a <= addr < a + size else ("ENDS AT" if a + size == addr else ("BEGINS AT" if a == addr else ""))
            print("segment %#x .. %#x  %8.2f MiB stream %s %s" % (a, a + size, size / 2**20, stream, rel))
            off = a
            for baddr, bsize, state in blocks:
                baddr = off if baddr is None else baddr
                if abs(baddr - addr) < (8 << 20) or abs(baddr + bsize - addr) < (8 << 20):
                    print("    block %#x .. %#x %10d B %s" % (baddr, baddr + bsize, bsize, state))
                off = baddr + bsize
    print("-- events within 64 MiB of the address (oldest first)")
    for action, a, size, stream, frames in d["events"]:
        if a is not None and abs(a - addr) < (64 << 20):
            print("%-18s %#x %10s stream %s  %s" % (action, a, size, stream, " < ".join(frames)))
