#!/bin/bash
# Round-3 GPU call A: wide Q|K|V kernel -- parity first, then A/B timing against the round-2 kernel, then the whole suite.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r3a
timeout 300 python -m pytest tests -m gpu -q -x -k "wide_typed_linear or typed_linear_against" 2>&1 | tail -30 > ${O}_pytest_linear.log
cat ${O}_pytest_linear.log | tail -15
for n in 768 512; do
  timeout 120 python tools/bench_linear.py --which bf16x3 --n-out $n 2>&1 | tail -2 | sed "s/^/wide n_out=$n: /"
  timeout 120 python tools/bench_linear.py --which bf16x3 --n-out $n --keep-pc 2>&1 | tail -2 | sed "s/^/pc   n_out=$n: /"
done | tee ${O}_linear_ab.log
if grep -q passed ${O}_pytest_linear.log && ! grep -q failed ${O}_pytest_linear.log; then
  timeout 200 python bench.py --no-cpu-baseline --no-secondary > ${O}_bench_wide.json 2> ${O}_bench_wide.err
  timeout 200 python bench.py --no-cpu-baseline --no-secondary --kernel-flags 4 > ${O}_bench_pc.json 2> ${O}_bench_pc.err
  for f in ${O}_bench_wide.json ${O}_bench_pc.json; do python - "$f" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], "ms", round(j["ms_per_step"],3), "parity", j["parity_max_abs_err"], j["roofline"]["phase_ms"])
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
  done
  timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -15 > ${O}_pytest_all.log
  tail -8 ${O}_pytest_all.log
fi
