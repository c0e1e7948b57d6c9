#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per (kernel, grid) calls, total, average,
min, max and share of GPU time.  Usage: rocpd_summary.py <results.db> [> profiles/xxx.txt]"""
import re
import sqlite3
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    m = re.match(r"void conv_igemm_kernel<(.*?)>\(ConvArgs\)", name)
    if m:
        return "conv_igemm<%s>" % m.group(1).replace(" ", "")
    return re.sub(r"\(.*\)$", "", name.replace("void ", ""))[:70]


def main(db):
    c = sqlite3.connect(db)
    rows = c.execute("select name, grid_x, grid_y, workgroup_x, lds_size, vgpr_count, accum_vgpr_count, "
                     "count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
                     "group by name, grid_x, grid_y order by sum(duration) desc").fetchall()
    tot = sum(r[8] for r in rows)
    print("# rocprofv3 --kernel-trace --stats summary (%s)" % db.split("/")[-1])
    print("# total kernel time %.3f ms over %d dispatches" % (tot / 1e6, sum(r[7] for r in rows)))
    print("%-58s %9s %5s %6s %5s %6s %10s %9s %9s %9s %6s" % ("kernel", "grid", "wg", "lds", "vgpr", "calls",
                                                            "total_us", "avg_us", "min_us", "max_us", "pct"))
    # per kernel over all grids: the average duration bench.py's roofline object is checked against
    agg = {}
    for r in rows:
        d = agg.setdefault(short(r[0]), [0, 0.0])
        d[0] += r[7]
        d[1] += r[8]
    print("# per kernel, all grids:  calls  total_us  avg_us  pct")
    for k, (n, s) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
        print("#   %-56s %7d %11.1f %9.2f %6.2f" % (k, n, s / 1e3, s / n / 1e3, 100.0 * s / tot))
    for r in rows:
        name, gx, gy, wx, lds, vg, ag, n, s, a, mn, mx = r
        print("%-58s %9s %5d %6d %5d %6d %10.1f %9.2f %9.2f %9.2f %6.2f" % (
            short(name), "%dx%d" % (gx // max(wx, 1), gy), wx, lds, vg + ag, n, s / 1e3, a / 1e3, mn / 1e3,
            mx / 1e3, 100.0 * s / tot))


if __name__ == "__main__":
    main(sys.argv[1])
