#!/bin/bash
# Collect rocprofv3 hardware counters for one conv_bench configuration, in separate passes
# (SQ: 8 slots per pass; FETCH_SIZE and WRITE_SIZE cannot share a pass -- MI355X_MICROARCH.md).
# usage: tools/pmc.sh <tag> <conv_bench args...>     -> gpurun_out/pmc_<tag>.txt
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
PASSES=(
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES"
  "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT"
  "FETCH_SIZE GRBM_GUI_ACTIVE"
  "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"
)
i=0
for P in "${PASSES[@]}"; do
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d "$OUT/p$i" -o p -- python "$R/tools/conv_bench.py" "$@" --reps 6 > "$OUT/p$i.log" 2>&1
  i=$((i+1))
done
python - "$OUT" > "$R/gpurun_out/pmc_$TAG.txt" <<'EOF'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "conv" not in k:
            continue
        agg[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    print("kernel", k)
    for c, v in sorted(cs.items()):
        print("  %-28s avg %.4g  (n=%d)" % (c, sum(v) / len(v), len(v)))
EOF
cat "$R/gpurun_out/pmc_$TAG.txt"
