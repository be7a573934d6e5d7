#!/bin/bash
# Round-3 validation + profiles of one build: pytest -m gpu, the judged bench line, rocprofv3 passes (kernel stats + counters) of
# the judged command, kernel stats of the latency-regime workloads.  Summaries land in gpurun_out/ (copied to profiles/ by hand).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export HGT_COMMIT=$(cat .commit 2>/dev/null || echo unknown)
bash tools/gpu_validate.sh r03
tools/profile_pmc.sh r03 > gpurun_out/prof_r03.log 2>&1
tail -3 gpurun_out/prof_r03.log
HGT_SMALL_PRECS=bf16x3,f16x3 bash tools/profile_small.sh r03 | tail -25
python tools/bench_train.py > gpurun_out/r03_bench_train.log 2>&1; tail -2 gpurun_out/r03_bench_train.log
