#!/bin/bash
# tools/lab/lab_run.sh <name> ...: the judged layer (c2, no secondaries, no parity gate) on pyhgt_amd/lib_lab_<name>/libhgt_hip.so
# ("prod" = pyhgt_amd/lib, "dev" = pyhgt_amd/lib_lab); prints ms per step and the phase split
cd ${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p gpurun_out
for n in "$@"; do
    case $n in prod) L=pyhgt_amd/lib/libhgt_hip.so;; dev) L=pyhgt_amd/lib_lab/libhgt_hip.so;; *) L=pyhgt_amd/lib_lab_$n/libhgt_hip.so;; esac
    HGT_LIB_PATH=$(pwd)/$L timeout 300 python bench.py --no-secondary --no-cpu-baseline --no-parity --steps ${STEPS:-30} $EXTRA_ARGS > gpurun_out/lab_$n.json 2> gpurun_out/lab_$n.err || tail -3 gpurun_out/lab_$n.err
    python - $n <<'PY'
import json, sys
n = sys.argv[1]
try:
    j = json.loads(open("gpurun_out/lab_%s.json" % n).read().strip().splitlines()[-1])
    print("%-10s ms %.3f  phases %s" % (n, j["ms_per_step"], j["roofline"].get("phase_ms")))
except Exception as e:
    print(n, "failed", e)
PY
done
