#!/bin/bash
# Full GPU validation of a build (run through gpurun from the repo root): pytest -m gpu, the judged bench line + RTE / Zipf variants,
# a 2-rank walk of the multi-GPU path on one device (gloo, host-staged exchange), the parity fuzzers, the rocprofv3 passes of
# tools/profile_pmc.sh, the latency-regime and training-step timings.  Writes gpurun_out/*; summaries are copied to profiles/ by hand.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export HGT_COMMIT=$(cat .commit 2>/dev/null || echo unknown)
python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/pytest_r2e.log
python bench.py > gpurun_out/bench_r2e.json 2> gpurun_out/bench_r2e.err
python bench.py --rte --no-cpu-baseline --no-secondary > gpurun_out/bench_r2e_rte.json 2>> gpurun_out/bench_r2e.err
python bench.py --dst-skew 0.8 --no-cpu-baseline --no-secondary > gpurun_out/bench_r2e_zipf.json 2>> gpurun_out/bench_r2e.err
HGT_BENCH_DEVICE=0 HGT_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 --nodes-per-gpu 200000 --edges-per-gpu 2000000 > gpurun_out/bench_r2e_dist2.json 2> gpurun_out/bench_r2e_dist2.err
python tools/fuzz_parity.py 80 > gpurun_out/fuzz_r2e.log 2>&1; tail -1 gpurun_out/fuzz_r2e.log; grep -c FAIL gpurun_out/fuzz_r2e.log
python tools/fuzz_parity.py 16 big > gpurun_out/fuzz_big_r2e.log 2>&1; tail -1 gpurun_out/fuzz_big_r2e.log
tools/profile_pmc.sh r02 > gpurun_out/prof_r02.log 2>&1
python tools/bench_small.py > gpurun_out/bench_small_r2e.log 2>&1; tail -1 gpurun_out/bench_small_r2e.log
python tools/bench_train.py > gpurun_out/bench_train_r2e.log 2>&1; tail -1 gpurun_out/bench_train_r2e.log
tail -12 gpurun_out/pytest_r2e.log
for f in gpurun_out/bench_r2e*.json; do python - "$f" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], "ms", round(j["ms_per_step"],3), "parity", j["parity_max_abs_err"], j["roofline"]["phase_ms"], {k:(round(v["ms_per_step"],3), v["parity_max_abs_err"]) for k,v in j.get("secondary",{}).items()}, (j.get("cpu_baseline") or {}).get("sample"))
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done
tail -c 600 gpurun_out/bench_r2e_dist2.err
cat gpurun_out/prof_r02/r02_pmc_summary.json | python -c "
import json,sys
j=json.load(sys.stdin)
for k in ('avg_kernel_ms','traffic_bytes','mfma_busy_pct','valu_busy_pct','waves_per_simd_avg','wait_pct','lds_bank_conflict_pct'): print(k, j.get(k))
"
