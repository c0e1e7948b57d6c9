#!/usr/bin/env python
"""How full is the GPU during a training step?  From a rocprofv3 (rocpd sqlite) kernel trace: for the longest busy burst (the
timed steps) the time covered by (a) at least one WIDE kernel -- >= 128 workgroups, i.e. one that can occupy the 256 CUs --,
(b) only narrow kernels (finalize / reduce / small elementwise launches: the GPU is "busy" but nearly empty), (c) nothing; and
per queue the busy time and the kernels that account for the narrow-only time.

  python tools/trace_util.py <results.db> [wide_workgroups]"""
import collections
import sqlite3
import sys


def union(iv):
    out = []
    for s, e in sorted(iv):
        if out and s <= out[-1][1]:
            out[-1][1] = max(out[-1][1], e)
        else:
            out.append([s, e])
    return out


def covered(iv):
    return sum(e - s for s, e in iv)


def subtract(a, b):
    """a minus b, both unions (sorted, disjoint)."""
    out, j = [], 0
    for s, e in a:
        cur = s
        while j < len(b) and b[j][1] <= cur:
            j += 1
        k = j
        while k < len(b) and b[k][0] < e:
            if b[k][0] > cur:
                out.append([cur, b[k][0]])
            cur = max(cur, b[k][1])
            k += 1
        if cur < e:
            out.append([cur, e])
    return out


def main(db, wide=128):
    c = sqlite3.connect(db)
    rows = c.execute("select start, end, grid_x * grid_y * grid_z / (workgroup_x * workgroup_y * workgroup_z), name, queue_id from kernels").fetchall()
    allu = union([(s, e) for s, e, *_ in rows])
    bursts, cur = [], [allu[0]]
    for m in allu[1:]:
        if m[0] - cur[-1][1] > 1e6:
            bursts.append(cur)
            cur = [m]
        else:
            cur.append(m)
    bursts.append(cur)
    import bisect
    starts = sorted(r[0] for r in rows)
    nk = lambda x: bisect.bisect_right(starts, x[-1][1]) - bisect.bisect_left(starts, x[0][0])
    b = max(bursts, key=nk)            # the timed training steps: the burst with the most kernels
    t0, t1 = b[0][0], b[-1][1]
    short = lambda n: n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:60]
    rows = [(s_, e, wg, short(n), q) for s_, e, wg, n, q in rows if t0 <= s_ <= t1]
    span = (t1 - t0) / 1e6
    wide_u = union([(s, e) for s, e, wg, *_ in rows if wg >= wide])
    all_u = union([(s, e) for s, e, *_ in rows])
    narrow_only = subtract(all_u, wide_u)
    print("burst %.1f ms, %d kernels: wide (>= %d workgroups) kernels cover %.1f %%, narrow kernels alone %.1f %%, idle %.1f %%"
          % (span, len(rows), wide, 100 * covered(wide_u) / 1e6 / span, 100 * covered(narrow_only) / 1e6 / span, 100 * (span - covered(all_u) / 1e6) / span))
    for q in sorted({r[4] for r in rows}):
        qu = union([(s, e) for s, e, wg, n, qq in rows if qq == q])
        print("  queue %d: %d kernels, busy %.1f %% of the burst" % (q, sum(1 for r in rows if r[4] == q), 100 * covered(qu) / 1e6 / span))
    # which narrow kernels own the narrow-only time
    acc = collections.Counter()
    cnt = collections.Counter()
    for s, e, wg, n, q in rows:
        if wg >= wide:
            continue
        own = covered(subtract([[s, e]], wide_u))
        acc[n] += own
        cnt[n] += 1
    print("  narrow kernels by time outside any wide kernel (ms over the burst, launches):")
    for n, t in acc.most_common(14):
        print("    %-62s %7.2f ms  %5d" % (n, t / 1e6, cnt[n]))
    # sweep: time with exactly one wide kernel resident (attributed to it) / with two or more (the two streams overlapping)
    ev = []
    for i, (s_, e, wg, n, q) in enumerate(rows):
        if wg >= wide:
            ev += [(s_, 1, i), (e, -1, i)]
    ev.sort()
    active, last, solo, multi = set(), None, collections.Counter(), 0
    for t, d, i in ev:
        if last is not None and active:
            if len(active) == 1:
                solo[rows[next(iter(active))][3]] += t - last
            else:
                multi += t - last
        last = t
        if d > 0:
            active.add(i)
        else:
            active.discard(i)
    print("  one wide kernel resident: %.1f %% of the burst; two or more: %.1f %%" % (100 * sum(solo.values()) / 1e6 / span, 100 * multi / 1e6 / span))
    print("  wide kernels by time as the ONLY wide kernel (ms over the burst):")
    for n, t in solo.most_common(16):
        print("    %-62s %7.2f ms" % (n, t / 1e6))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 128))
