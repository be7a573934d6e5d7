#!/bin/bash
# one-off GPU script of the x-stationary GEMM experiment (run through gpurun from the repo root)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
ulimit -c 0
timeout 420 python tools/bench_xs.py > gpurun_out/xs_check.log 2>&1; echo "bench_xs rc=$?"
tail -40 gpurun_out/xs_check.log
for m in 0 1; do
  HGT_GEMM_XS=$m timeout 300 python bench.py --no-secondary --no-cpu-baseline > gpurun_out/xs_bench_$m.json 2> gpurun_out/xs_bench_$m.err; echo "bench XS=$m rc=$?"
  python - <<PY
import json
try:
    j = json.loads(open("gpurun_out/xs_bench_$m.json").read().strip().splitlines()[-1])
    r = j.get("roofline", {})
    print("XS=$m ms", round(j["ms_per_step"], 3), "parity", j.get("parity_max_abs_err"), "layer_frac", r.get("layer_frac"), r.get("phase_ms"))
except Exception as e:
    print("unreadable", e)
PY
done
HGT_GEMM_XS=1 timeout 600 python -m pytest tests -m gpu -q -x -k "xs_gemm or typed_linear or matches_oracle or golden or 24_bit or real_halos or target_block" > gpurun_out/xs_pytest.log 2>&1; echo "pytest rc=$?"
grep -E "^FAILED|^ERROR| passed| failed" gpurun_out/xs_pytest.log | tail -8; grep -E "^E  " gpurun_out/xs_pytest.log | head -10
