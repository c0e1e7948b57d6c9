#!/usr/bin/env python
"""GPU idle time between kernels in a rocprofv3 (rocpd sqlite) kernel trace: how much of a training step is launch gaps?

  python tools/trace_gaps.py <results.db> [min_busy_ms]
Takes the union of all kernels' [start, end] intervals (every queue), splits the trace into bursts at idle periods longer than
1 ms (host synchronisations between the bench's phases), and for the bursts longer than min_busy_ms (default 100: the timed
steps) reports busy time, idle time inside the burst, the number of gaps and their size distribution."""
import sqlite3
import sys


def main(db, min_ms=100.0):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    s_col = "start" if "start" in cols else "start_timestamp"
    e_col = "end" if "end" in cols else "end_timestamp"
    iv = sorted(c.execute("select %s, %s from kernels" % (s_col, e_col)).fetchall())
    merged = []
    for s, e in iv:
        if merged and s <= merged[-1][1]:
            merged[-1][1] = max(merged[-1][1], e)
        else:
            merged.append([s, e])
    bursts, cur = [], [merged[0]]
    for m in merged[1:]:
        if m[0] - cur[-1][1] > 1e6:
            bursts.append(cur)
            cur = [m]
        else:
            cur.append(m)
    bursts.append(cur)
    for b in bursts:
        span = (b[-1][1] - b[0][0]) / 1e6
        if span < min_ms:
            continue
        busy = sum(e - s for s, e in b) / 1e6
        gaps = [b[i + 1][0] - b[i][1] for i in range(len(b) - 1)]
        nk = sum(1 for s, e in iv if b[0][0] <= s <= b[-1][1])
        hist = [sum(1 for g in gaps if lo <= g < hi) for lo, hi in ((0, 1e3), (1e3, 3e3), (3e3, 1e4), (1e4, 1e5), (1e5, 1e9))]
        print("burst %.1f ms: %d kernels, busy (union) %.1f ms, idle %.2f ms = %.1f %%, %d gaps; <1us %d, 1-3us %d, 3-10us %d, 10-100us %d, >100us %d; "
              "idle in gaps <10us: %.2f ms" % (span, nk, busy, span - busy, 100 * (span - busy) / span, len(gaps), *hist,
                                             sum(g for g in gaps if g < 1e4) / 1e6))


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 100.0)
