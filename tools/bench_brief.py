"""Print the headline numbers of a bench.py JSON line: python tools/bench_brief.py <file>"""
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("step %.3f ms  %.1f img/s  launches %s" % (d["ms_per_step"], d["value"], d["config"].get("library_launches_per_step")))
r = d["roofline"]
print("dominant %s frac %.3f (%.0f GB/s, %.1f us)" % (r["kernel"], r["frac"], r["achieved"], r["avg_launch_us"]))
print("ranking", r.get("ranking_ms_per_step"))
f = d.get("forward_only")
if f:
    print("forward %.3f ms  %.0f img/s  headline %s" % (f["ms_per_step"], f["value"], f["roofline"].get("headline_3x3")))
if d.get("forward_1024"):
    print("forward_1024", d["forward_1024"]["value"], d["forward_1024"]["ms_per_step"])
if d.get("freqsplit_1024"):
    q = d["freqsplit_1024"]
    print("freqsplit_1024", {k: (v["us"], v["frac_hbm"]) for k, v in q.items() if isinstance(v, dict)})
print("losses", d["config"].get("last_losses"))
