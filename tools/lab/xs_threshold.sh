#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
timeout 300 python tools/bench_xs.py --threshold 2>&1 | grep -v amdgpu.ids | tee gpurun_out/xs_threshold.txt
