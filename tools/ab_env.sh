#!/bin/bash
# A/B of environment switches on the training-step bench: tools/ab_env.sh "VAR=1" "VAR=2 OTHER=x" ...  (each run: bench.py
# --no-forward-1024 with that environment; prints ms/step).  The first run is the unmodified one; it is repeated at the end
# so that box drift shows.
set -u
run() { env $1 python bench.py --no-forward-1024 --no-cpu-baseline --no-forward-leg --steps ${STEPS:-30} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-60s %7.3f ms  %6.1f images/s  frac %.3f  host enqueue %.2f ms (in-loop %.2f)  launches %d' % (sys.argv[1], d['ms_per_step'], d['value'], d['roofline']['frac'], d['config']['host_enqueue_ms_per_step'], d['config']['host_enqueue_ms_per_step_backpressured'], d['config']['library_launches_per_step']))" "$1"; }
run "_=base"
for e in "$@"; do run "$e"; done
run "_=base"
