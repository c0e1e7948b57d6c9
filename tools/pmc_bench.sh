#!/bin/bash
# HBM traffic per kernel of the benchmark step, from rocprofv3 hardware counters (separate passes:
# FETCH_SIZE and WRITE_SIZE cannot share one -- MI355X_MICROARCH.md).  Writes profiles-ready JSON:
#   tools/pmc_bench.sh  ->  gpurun_out/pmc_bench.json  (copy to profiles/pmc_traffic.json)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/pmc_bench
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for P in "FETCH_SIZE" "WRITE_SIZE"; do
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d "$OUT/p$i" -o p -- python "$R/bench.py" --no-cpu-baseline --no-forward-1024 --steps 2 --warmup 1 "$@" > "$OUT/p$i.log" 2>&1
  i=$((i+1))
done
python - "$OUT" "$R/gpurun_out/pmc_bench.json" <<'PY'
import collections, csv, glob, json, sys
out, dst = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
# kernel symbol -> (launcher name, workload key): the forward kernels are keyed on the forward-only workload, the
# BatchNorm-backward pass on the training step (both run in the default bench.py)
names = {"conv1x1_ds_kernel": ("conv1x1_ds_bn128", "netG_B16_256"), "conv3x3_rs2_kernel": ("conv3x3_rs2_bn32", "netG_B16_256"),
         "bn_bwd_apply_kernel": ("bn_bwd_apply", "train_B16_256"),
         "conv1x1_bwd_kernel": ("conv1x1_bwd_stream", "train_B16_256"),
         "conv1x1_bwdw_kernel": ("conv1x1_bwd_wgrad_stream", "train_B16_256"),
         "conv3x3_bwd_kernel": ("conv3x3_bwd_stream", "train_B16_256"),
         "conv3x3_bwd2_kernel": ("conv3x3_bwd_stream2", "train_B16_256"),
         "conv_wgrad_r3_kernel": ("conv_wgrad3x3_r3", "train_B16_256"),
         "conv_wgrad_r4_kernel": ("conv_wgrad4x4_r4", "train_B16_256"),
         "affine_acc_kernel": ("affine_accumulate", "train_B16_256"),
         "conv_wgrad1x1_tr_kernel": ("conv_wgrad1x1_tr", "train_B16_256"),
         "conv_wgrad_tr_kernel<3, 3, 8>": ("conv_wgrad3x3_tr8", "train_B16_256"),
         "conv_wgrad_tr_kernel<4, 2, 9>": ("conv_wgrad4x4_tr", "train_B16_256"),
         "conv_igemm_kernel<3, 1, 0, 8, 2, 1, 4, 9, 0, 1>": ("conv3x3_wd128", "train_B16_256"),
         "conv_igemm_kernel<3, 1, 0, 8, 2, 1, 4, 9, 1, 1>": ("conv3x3_wd128_bwd", "train_B16_256"),
         "conv_igemm_kernel<4, 1, 0, 8, 3, 1, 3, 16, 0, 1>": ("conv4x4_wd144", "train_B16_256"),
         "conv_igemm_kernel<4, 1, 0, 8, 3, 1, 3, 16, 1, 1>": ("conv4x4_wd144_bwd", "train_B16_256")}
res = {"_comment": "average per launch over every launch of the kernel in `python bench.py --no-forward-1024` (training step + netG forward leg, B=16 @256^2 only); KiB; "
                   "FETCH_SIZE is x2-corrected by bench.py per MI355X_MICROARCH.md (gfx950 reports half of wide coalesced reads)"}
for k, cs in agg.items():
    for sym, (nm, wl) in names.items():
        if sym in k and "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
            res["%s@%s" % (nm, wl)] = {"fetch_kib": sum(cs["FETCH_SIZE"]) / len(cs["FETCH_SIZE"]),
                                           "write_kib": sum(cs["WRITE_SIZE"]) / len(cs["WRITE_SIZE"]),
                                           "launches": len(cs["FETCH_SIZE"]), "symbol": k[:80]}
json.dump(res, open(dst, "w"), indent=1)
print(json.dumps(res, indent=1))
PY
