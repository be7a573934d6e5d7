#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 300 python bench.py --no-secondary --no-cpu-baseline > gpurun_out/sanity_bench.json 2> gpurun_out/sanity_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
j = json.loads(open("gpurun_out/sanity_bench.json").read().strip().splitlines()[-1])
r = j["roofline"]
print("ms", round(j["ms_per_step"], 3), "parity", j["parity_max_abs_err"], "frac", r["frac"], "layer_frac", r["layer_frac"], r["phase_ms"])
PY
timeout 300 python -m pytest tests -m gpu -q -x -k "golden or xs_gemm or reference_call" 2>&1 | tail -2
