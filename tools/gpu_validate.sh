#!/bin/bash
# GPU validation of a build (run through gpurun from the repo root): pytest -m gpu, then the judged bench line.
#   tools/gpu_validate.sh <tag> [pytest -k expression]
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=${1:-r3}; K=${2:-}
if [ -n "$K" ]; then
  timeout 1500 python -m pytest tests -m gpu -q -x -k "$K" -s 2>&1 | grep -v "^$" | tail -60 > gpurun_out/${TAG}_pytest.log
else
  timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 | grep -E "passed|failed|error|Error|err |error |worst|handoff|configs\[2\]|training path|assert" | tail -80 > gpurun_out/${TAG}_pytest.log
fi
tail -40 gpurun_out/${TAG}_pytest.log
( time timeout 600 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err ) 2>&1 | grep real
tail -c 600 gpurun_out/${TAG}_bench.err
python - gpurun_out/${TAG}_bench.json <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("ms", round(j["ms_per_step"],3), "parity", j["parity_max_abs_err"], j["roofline"]["phase_ms"], "stale", j["roofline"].get("pmc_stale"))
    for k,v in j.get("secondary",{}).items():
        if k=="latency_regime":
            for kk,vv in v.items():
                print(" ", kk, {a:(round(b.get("us_per_layer",b.get("us_per_forward",0)),1), "%.1e"%b["parity_max_abs_err"]) for a,b in vv.items() if isinstance(b,dict)}, {a:round(b,1) for a,b in vv.items() if isinstance(b,float)})
        else:
            print(" ", k, round(v["ms_per_step"],3), v.get("layer_frac"), "parity %.1e" % v["parity_max_abs_err"] if v.get("parity_max_abs_err") is not None else None, v.get("phase_ms"))
    print(" cpu", (j.get("cpu_baseline") or {}).get("sample"))
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
