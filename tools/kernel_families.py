"""Per-kernel-family time per step from a rocpd_summary.py listing: kernel_families.py <summary.txt> <steps>"""
import collections, re, sys
rows = [l for l in open(sys.argv[1]) if not l.startswith("#") and not l.startswith("kernel")]
steps = float(sys.argv[2])
agg = collections.defaultdict(lambda: [0, 0.0])
for l in rows:
    f = l.split()
    name = re.sub(r"<.*", "", " ".join(f[:-10]))
    agg[name][0] += int(f[-6]); agg[name][1] += float(f[-5])
T = sum(v[1] for v in agg.values())
print("kernel time %.2f ms/step" % (T / steps / 1e3))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print("%-52s %6.1f launches %8.2f ms/step %5.1f%%" % (k[:52], v[0] / steps, v[1] / steps / 1e3, 100 * v[1] / T))
