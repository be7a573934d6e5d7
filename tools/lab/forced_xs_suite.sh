#!/bin/bash
# the whole GPU suite with the x-stationary GEMM forced onto every eligible typed linear (its size threshold keeps it off small graphs)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
HGT_GEMM_XS=1 timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 | tee gpurun_out/forced_xs_suite.txt
