#!/bin/bash
# rocprofv3 kernel trace of the default bench.py workload -> gpurun_out/<tag>_kernel_stats.txt (copy to profiles/).
# usage: tools/prof_bench.sh <tag> [bench.py args]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT" -o p -- python "$R/bench.py" --no-cpu-baseline --steps 20 --warmup 3 "$@" > "$OUT/bench.log" 2>&1
DB=$(find "$OUT" -name "*.db" | head -1)
python "$R/tools/rocpd_summary.py" "$DB" > "$R/gpurun_out/${TAG}_kernel_stats.txt"
grep "^{" "$OUT/bench.log" | tail -n 1 > "$R/gpurun_out/${TAG}_prof_bench.json"
find "$OUT" -name "*.db" -delete
head -45 "$R/gpurun_out/${TAG}_kernel_stats.txt"
