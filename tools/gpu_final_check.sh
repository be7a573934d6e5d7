#!/bin/bash
# last call of a round: smoke(), the example training loop, the judged line (stored), counters of the same build
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export HGT_COMMIT=$(cat .commit 2>/dev/null || echo unknown)
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 300 python examples/train_synthetic.py --steps 6 2>&1 | tail -3
tools/profile_pmc.sh r03 > gpurun_out/prof_r03.log 2>&1
cp gpurun_out/prof_r03/r03_pmc_summary.json profiles/r03_pmc_summary.json      # so that the line below is judged against this build's counters
( time timeout 600 python bench.py > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err ) 2>&1 | grep real
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r03_bench.json").read().strip().splitlines()[-1])
print("ms", round(j["ms_per_step"], 3), "parity", j["parity_max_abs_err"], "frac", j["roofline"]["frac"], "layer_frac", j["roofline"]["layer_frac"], "stale", j["roofline"]["pmc_stale"], "traffic", j["roofline"]["traffic"])
PY
