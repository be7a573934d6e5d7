#!/bin/bash
# tools/lab/xs_nw4.sh: the x-stationary GEMM with four-wavefront workgroups (two per CU) for K = 256 against the shipped eight-wavefront form
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for L in pyhgt_amd/lib/libhgt_hip.so pyhgt_amd/lib_lab_nw4/libhgt_hip.so; do
    echo "== $L"
    HGT_LIB_PATH=$PWD/$L timeout 300 python tools/bench_xs.py --quick 2>&1 | grep -E "ALL|DIFFER|k=256 n_out=768" | tail -6
    for p in f16x3 bf16x3; do
    HGT_LIB_PATH=$PWD/$L timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 20 --precision $p 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('$p', round(j['ms_per_step'],3), 'parity', j.get('parity_max_abs_err'), 'layer_frac', r.get('layer_frac'), r.get('phase_ms'))"
    done
done
