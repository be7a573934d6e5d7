#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
timeout 200 python tools/lab/stream_ceiling.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/stream_ceiling.txt
