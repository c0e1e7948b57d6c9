#!/bin/bash
# One GPU call that regenerates the round's profile set under gpurun_out/ (copy the summaries to profiles/):
#   tools/refresh_profiles.sh r3_v1
TAG=${1:-r4}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
tools/prof_bench.sh $TAG --no-forward-1024 > gpurun_out/${TAG}_prof.log 2>&1
tools/pmc_bench.sh > gpurun_out/${TAG}_pmc_traffic.log 2>&1 && cp gpurun_out/pmc_bench.json gpurun_out/${TAG}_pmc_traffic.json
tools/pmc_mfma.sh > gpurun_out/${TAG}_pmc_mfma.log 2>&1 && cp gpurun_out/pmc_mfma.json gpurun_out/${TAG}_pmc_mfma_train_step.json
python bench.py --breakdown gpurun_out/${TAG}_step_breakdown.json > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python tools/bench_brief.py gpurun_out/${TAG}_bench.json
# power / clock evidence: the whole step, then the hot kernels one at a time (launch replayed for 4 s each)
python tools/step_power.py 8 > gpurun_out/${TAG}_power_step.txt 2>&1
python tools/power_probe.py --seconds 4 -- --k 3 --cin 128 --cout 32 --n 16 --hw 256 --bn --stats --pitch-out 256 > gpurun_out/${TAG}_power_conv3x3_rs2_headline.txt 2>&1
python tools/power_probe.py --seconds 4 -- --k 1 --cin 256 --cout 128 --n 16 --hw 256 --bn --stats > gpurun_out/${TAG}_power_conv1x1_ds.txt 2>&1
python tools/power_probe.py --seconds 4 -- --k 3 --cin 160 --cout 128 --n 16 --hw 128 > gpurun_out/${TAG}_power_conv3x3_wd128.txt 2>&1
tail -3 gpurun_out/${TAG}_power_*.txt
