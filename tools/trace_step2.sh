#!/bin/bash
# rocprofv3 kernel trace of a few training steps -> gpurun_out/<tag>_step.tsv (tools/trace_dump.py: one step, every kernel with queue / start / duration)
set -u
TAG=${1:-step}; shift || true
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/trace_$TAG
mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace -d "$OUT" -o p -- python "$R/bench.py" --no-cpu-baseline --no-forward-leg --no-forward-1024 --steps 6 --warmup 3 "$@" > "$OUT/bench.log" 2>&1
DB=$(find "$OUT" -name "*.db" | head -1)
python "$R/tools/trace_dump.py" "$DB" > "$R/gpurun_out/${TAG}_step.tsv" 2> "$R/gpurun_out/${TAG}_step.err"
cat "$R/gpurun_out/${TAG}_step.err"
python "$R/tools/trace_util.py" "$DB" > "$R/gpurun_out/${TAG}_trace_util.txt" 2>&1
rm -rf "$OUT"
