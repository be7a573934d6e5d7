#!/bin/bash
# tools/lab/d512_ab.sh "<flags a>" "<flags b>" ...: the d = 512 / d = 400 secondary workloads (N = 500 k, E = 5 M, 8 heads) under kernel-flag sets
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for d in 512 400; do for fl in "$@"; do
    echo "== d $d flags $fl"
    timeout 300 python bench.py --nodes-per-gpu 500000 --edges-per-gpu 5000000 --dim $d --no-cpu-baseline --no-secondary --steps 10 --warmup 3 --kernel-flags $fl 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print(round(j['ms_per_step'],3), 'parity', j.get('parity_max_abs_err'), 'layer_frac', r.get('layer_frac'), r.get('phase_ms'))"
done; done
