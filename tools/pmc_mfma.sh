#!/bin/bash
# MFMA-utilisation / LDS-conflict counters of the training step's kernels (north_star: "rocprof HBM GB/s and MFMA utilisation"):
# one SQ pass over the default bench.py workload -> gpurun_out/pmc_mfma.json (copy to profiles/).  Counters only with
# --kernel-trace (no other trace domain), as the pool requires.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/pmc_mfma
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT \
  --kernel-trace --output-format csv -d "$OUT/p0" -o p -- python "$R/bench.py" --no-cpu-baseline --no-forward-leg --steps 2 --warmup 1 > "$OUT/p0.log" 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/p1" -o p -- python "$R/bench.py" --no-cpu-baseline --no-forward-leg --steps 2 --warmup 1 > "$OUT/p1.log" 2>&1
python - "$OUT" "$R/gpurun_out/pmc_mfma.json" <<'PY'
import collections, csv, glob, json, re, sys
out, dst = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*\)$", "", r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", ""))
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {"_comment": "per kernel, average per launch over the launches of `python bench.py --steps 2` (training step, B=16 @256^2). "
                   "mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x GRBM_GUI_ACTIVE / 8 XCDs): share of all SIMD-cycles "
                   "of the launch with the matrix pipe busy (16 cycles per v_mfma_f32_16x16x32_bf16); SQ_WAVE_CYCLES / SQ_WAIT_* are "
                   "quad-cycles summed over waves (MI355X_MICROARCH.md); lds_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS"}
rows = []
for k, cs in agg.items():
    if "SQ_INSTS_MFMA" not in cs:
        continue
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    if m.get("SQ_INSTS_MFMA", 0) < 1000:
        continue
    gui = m.get("GRBM_GUI_ACTIVE", 0.0)
    e = {"kernel": k[:90], "launches": len(cs["SQ_INSTS_MFMA"]), "mfma_insts": m["SQ_INSTS_MFMA"], "valu_insts": m.get("SQ_INSTS_VALU"),
         "mfma_busy_cycles": m.get("SQ_VALU_MFMA_BUSY_CYCLES"), "gui_active": gui,
         "mfma_busy_frac": round(m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * gui / 8.0), 4) if gui else None,
         "wait_any_frac": round(m.get("SQ_WAIT_ANY", 0.0) / m["SQ_WAVE_CYCLES"], 3) if m.get("SQ_WAVE_CYCLES") else None,
         "wait_inst_frac": round(m.get("SQ_WAIT_INST_ANY", 0.0) / m["SQ_WAVE_CYCLES"], 3) if m.get("SQ_WAVE_CYCLES") else None,
         "lds_bank_conflict_cycles": m.get("SQ_LDS_BANK_CONFLICT"),
         "lds_conflict_frac": round(m["SQ_LDS_BANK_CONFLICT"] / m["SQ_ACTIVE_INST_LDS"], 4) if m.get("SQ_ACTIVE_INST_LDS") else None}
    rows.append(e)
rows.sort(key=lambda e: -(e["mfma_busy_cycles"] or 0) * e["launches"])
res["kernels"] = rows
json.dump(res, open(dst, "w"), indent=1)
for e in rows[:16]:
    print("%-72s n=%4d mfma_busy %.3f wait %.2f/%.2f lds_conf %s" % (e["kernel"][:72], e["launches"], e["mfma_busy_frac"] or 0, e["wait_any_frac"] or 0, e["wait_inst_frac"] or 0, e["lds_conflict_frac"]))
PY
