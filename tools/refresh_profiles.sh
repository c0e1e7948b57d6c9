cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
tools/prof_bench.sh r2_v9 > gpurun_out/r2_v9_prof.log 2>&1
tools/pmc_bench.sh > gpurun_out/r2_v9_pmc_traffic.log 2>&1
tools/pmc_mfma.sh > gpurun_out/r2_v9_pmc_mfma.log 2>&1
python bench.py --breakdown gpurun_out/r2_v9_breakdown.json > gpurun_out/r2_v9_bench.json 2> gpurun_out/r2_v9_bench.err
python tools/bench_brief.py gpurun_out/r2_v9_bench.json
ls -la gpurun_out | head -40
