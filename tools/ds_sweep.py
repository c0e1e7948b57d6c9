#!/usr/bin/env python
"""conv1x1_ds (the dense layers' 1x1 bottleneck, forward) over the netG shapes, optionally with phase skips of a tuning build:
  FDGAN_LIB=fd-gan_amd/fdgan_hip/variants/libfdgan_hip_tune.so python tools/ds_sweep.py [phase masks ...]
One subprocess per mask (the library reads FDGAN_DEBUG_PHASES once).  Prints us / algorithmic GB/s per shape and mask."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [(256, 256, c) for c in (64, 96, 128, 160, 192, 224)] + [(128, 512, c) for c in (128, 224, 256, 352, 480)] + \
         [(64, 1024, c) for c in (256, 416, 512, 736, 992)]

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path[:0] = [ROOT, os.path.join(ROOT, "fd-gan_amd"), os.path.join(ROOT, "tools")]
    import conv_bench
    for hw, pitch, cin in SHAPES:
        r = conv_bench.run(1, cin, 128, 16, hw, hw, bn=True, stats=True, pitch_in=pitch, reps=12)
        print(json.dumps({"hw": hw, "cin": cin, "us": r["us"], "GB/s": r["GB/s"], "kernel": r["kernel"]}), flush=True)
    sys.exit(0)

masks = sys.argv[1:] or ["0"]
table = {}
for m in masks:
    env = dict(os.environ, FDGAN_DEBUG_PHASES=m)
    out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True)
    for ln in out.stdout.splitlines():
        if ln.startswith("{"):
            d = json.loads(ln)
            table.setdefault((d["hw"], d["cin"]), {})[m] = (d["us"], d["GB/s"])
    if out.returncode:
        print(out.stderr[-2000:])
print("%-12s" % "shape" + "".join("%18s" % ("mask " + m) for m in masks))
for (hw, cin), row in sorted(table.items(), key=lambda kv: (-kv[0][0], kv[0][1])):
    print("%4d^2 c%-5d" % (hw, cin) + "".join("%9.1fus %5.0fGB/s" % row.get(m, (0, 0)) for m in masks))
