#!/bin/bash
# rocprofv3 kernel trace of a few training steps, analysed on the box: tools/trace_gaps.py (idle between kernels) and
# tools/trace_util.py (time covered by wide kernels / by narrow kernels alone / per queue) -> gpurun_out/<tag>_trace_*.txt
set -u
TAG=${1:-step}; shift || true
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/trace_$TAG
mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace -d "$OUT" -o p -- python "$R/bench.py" --no-cpu-baseline --no-forward-leg --steps 6 --warmup 3 "$@" > "$OUT/bench.log" 2>&1
DB=$(find "$OUT" -name "*.db" | head -1)
python "$R/tools/trace_gaps.py" "$DB" 20 > "$R/gpurun_out/${TAG}_trace_gaps.txt" 2>&1
python "$R/tools/trace_util.py" "$DB" > "$R/gpurun_out/${TAG}_trace_util.txt" 2>&1
cat "$R/gpurun_out/${TAG}_trace_util.txt"
rm -rf "$OUT"/*/  # the database stays on the box
