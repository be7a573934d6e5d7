#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
timeout 300 python tools/bench_xs.py 2>&1 | grep -v "BIT-IDENTICAL (\|amdgpu.ids" | tee gpurun_out/xs_quick.txt
timeout 200 python -m pytest tests -m gpu -q -k "xs_gemm or typed_linear" 2>&1 | tail -2
