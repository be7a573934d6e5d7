#!/bin/bash
# one-off GPU script of the x-stationary GEMM experiments (run through gpurun from the repo root)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
ulimit -c 0
timeout 420 python tools/bench_xs.py > gpurun_out/xs_check.log 2>&1; echo "bench_xs rc=$?"
grep -v "BIT-IDENTICAL (" gpurun_out/xs_check.log | tail -30
for m in 0 1; do
  HGT_GEMM_XS=$m timeout 300 python bench.py --dim 512 --nodes-per-gpu 500000 --edges-per-gpu 5000000 --no-secondary --no-cpu-baseline > gpurun_out/xs_d512_$m.json 2> gpurun_out/xs_d512_$m.err; echo "bench d512 XS=$m rc=$?"
  python - <<PY
import json
try:
    j = json.loads(open("gpurun_out/xs_d512_$m.json").read().strip().splitlines()[-1])
    r = j.get("roofline", {})
    print("d512 XS=$m ms", round(j["ms_per_step"], 3), "parity", j.get("parity_max_abs_err"), "layer_frac", r.get("layer_frac"), r.get("phase_ms"))
except Exception as e:
    print("unreadable", e)
PY
done
