#!/usr/bin/env python
"""One training step of a rocprofv3 (rocpd sqlite) kernel trace as a table: per kernel its queue, start offset, duration and name, for
the LAST complete step of the longest busy burst (a step = from one generator adam_kernel to the next).  For reading the critical
path by eye / by tools/trace_phases.py.     python tools/trace_dump.py <results.db> > step.tsv"""
import sqlite3
import sys


def main(db):
    c = sqlite3.connect(db)
    tables = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    kt = [t for t in tables if t.startswith("kernels")] or [t for t in tables if "kernel" in t]
    rows = c.execute("select start, end, name, queue_id, grid_x * grid_y * grid_z / (workgroup_x * workgroup_y * workgroup_z) from %s order by start" % kt[0]).fetchall()
    adam = [i for i, r in enumerate(rows) if "adam" in r[2]]
    # the generator's Adam is the last adam launch of a step (D's comes earlier): steps end at every second adam launch
    ends = adam[1::2]
    if len(ends) < 3:
        raise SystemExit("fewer than 3 steps in the trace")
    lo, hi = ends[-2] + 1, ends[-1] + 1
    t0 = rows[lo][0]
    queues = {}
    for s, e, name, q, wgs in rows[lo:hi]:
        qi = queues.setdefault(q, len(queues))
        print("%d\t%.1f\t%.1f\t%d\t%s" % (qi, (s - t0) / 1e3, (e - s) / 1e3, wgs, name[:70]))
    print("# step %.3f ms, %d kernels, queues %s" % ((rows[hi - 1][1] - t0) / 1e6, hi - lo, queues), file=sys.stderr)


if __name__ == "__main__":
    main(sys.argv[1])
