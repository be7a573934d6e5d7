#!/bin/bash
# Round-3 GPU call B: elimination / variant timings of the wide Q|K|V kernel (lab build, HGT_WD_VARIANT=n; see
# tools/lab/hgt_gemm_wide_lab.hip).  Timing only for the variants that skip work.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export HGT_LIB_PATH=$GRAFT_REPO_ROOT/pyhgt_amd/lib_lab/libhgt_hip.so
for v in "$@"; do
  HGT_WD_VARIANT=$v timeout 120 python tools/bench_linear.py --which bf16x3 --iters 20 2>&1 | tail -1 | sed "s/^/v$v: /"
done | tee gpurun_out/r3b_variants.log
env -u HGT_WD_VARIANT timeout 120 python tools/bench_linear.py --which bf16x3 --iters 20 2>&1 | tail -1 | sed "s/^/pc: /" | tee -a gpurun_out/r3b_variants.log
