#!/bin/bash
# PMC passes over one command, per-kernel averages: tools/pmc_one.sh <kernel-substring> <cmd...>   (GPU box)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
K=$1; shift
export TMPDIR=/tmp
PASSES=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT"
        "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC"
        "SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INST_CYCLES_SALU")
i=0
for P in "${PASSES[@]}"; do
  OUT=$(mktemp -d /tmp/pmcone.XXXX)
  ( cd /tmp && rocprofv3 --pmc $P --kernel-trace --output-format csv -d "$OUT" -o p -- "$@" > "$OUT/log" 2>&1 )
  python - "$OUT" "$K" <<'PY'
import collections, csv, glob, sys
out, key = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if key in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
if not agg:
    print("  (no counters: " + open(out + "/log").read()[-300:].replace("\n", " | ") + ")")
for c, v in sorted(agg.items()):
    print("  %-28s %14.0f  (%d launches)" % (c, sum(v) / len(v), len(v)))
PY
  rm -rf "$OUT"
  i=$((i+1))
done
