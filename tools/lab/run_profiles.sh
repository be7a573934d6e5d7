cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; ulimit -c 0
export HGT_COMMIT=$(cat .commit 2>/dev/null || echo unknown)
bash tools/gpu.sh profile r04 2>&1 | tail -12
bash tools/profile_train.sh r04 2>&1 | tail -25
( time timeout 1700 python bench.py --cpu-baseline-full --no-secondary > gpurun_out/bench_r04_cpufull.json 2> gpurun_out/bench_r04_cpufull.err ) 2>&1 | grep real
python - <<'PY'
import json
j = json.loads(open("gpurun_out/bench_r04_cpufull.json").read().strip().splitlines()[-1])
print(json.dumps(j["cpu_baseline"])[:1500])
print("ms", j["ms_per_step"], j["roofline"]["phase_ms"])
PY
for p in f16x3; do timeout 300 python bench.py --precision $p --no-secondary --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('f16x3 ms', j['ms_per_step'], j['roofline']['phase_ms'], j['parity_max_abs_err'])"; done
