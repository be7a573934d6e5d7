#!/bin/bash
# The one GPU-side driver script (run through gpurun from the repo root):   tools/gpu.sh <mode> [tag]
#   quick     the round's new / changed tests, micro-benchmarks of the typed linears (incl. the 24-bit wire source) and the emulated
#             rank-of-8 step of the multi-GPU path: minutes
#   validate  pytest -m gpu, the judged bench line + a 2-rank walk of the multi-GPU path on one device, fuzzers, small / training benches
#   profile   rocprofv3 passes of the judged command (tools/profile_pmc.sh <tag>) + kernel statistics of the latency regime
#   final     smoke(), the example training loop, pytest -m gpu, profile, the emulated rank-of-8 steps, then the judged line against this
#             build's counters
#   emuprof   rocprofv3 kernel statistics of the emulated rank-of-8 step (bench.py --emulate-world 8): the per-kernel split quoted in
#             DESIGN.md section 6
#   judged    rocprofv3 passes of the judged command + the judged line (what `final` ends with)
#   xs        the x-stationary typed linear (csrc/hgt_gemm_xs.hip): bit-identity against the slab kernel in every wavefront order,
#             timings with the elimination switches (tools/bench_xs.py), the c2 / d = 512 layers with and without it
#   suite     pytest -m gpu, whole suite ($3 = HGT_FLAG_* bits forced onto every layer through HGT_TEST_KERNEL_FLAGS, e.g. 1024 = the
#             x-stationary GEMM on every eligible typed linear, 512 = the LDS-ring aggregation)
#   sanity    smoke(), the judged layer without secondaries, a handful of tests: the first call after a kernel change
#   ceiling   plain streaming kernels of the part (tools/lab/stream_ceiling.py): fill / copy / 1:3 read:write rates
#   xsthr     slab vs x-stationary GEMM around the dispatch threshold (tools/bench_xs.py --threshold)
#   latency   kernel timelines of the latency-regime workloads (c3, script-default batch, c5, published 4-layer model, c1; $3 = precision):
#             per-kernel us and gaps of a steady-state forward (tools/trace_latency.py) -> gpurun_out/<tag>_latency_<workload>.txt
#   small     the latency-regime part of the judged line only (bench.py --small-only): wall-clock us per layer / forward + parity
# Everything lands in gpurun_out/; summaries to be judged are copied to profiles/ by hand (or by `final`).
# Kernel experiments: tools/lab/build_lab.sh (lab libraries), lab_run.sh (phase times), pmc_quick.sh (counters), isa.sh (ISA + resources).
MODE=${1:-quick}; TAG=${2:-r06}
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
ulimit -c 0      # a faulting kernel must not fill the scratch disk with core files (every later command of the call then fails)
export HGT_COMMIT=$(cat .commit 2>/dev/null || echo unknown)
summ() { python - "$@" <<'PY'
import json, sys
for f in sys.argv[1:]:
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    if "ms_per_step" not in j:
        print(f, json.dumps(j)[:900]); continue
    r = j.get("roofline", {})
    print(f, "ms", round(j["ms_per_step"], 3), "parity", j.get("parity_max_abs_err"), "frac", r.get("frac"), "layer_frac", r.get("layer_frac"),
          r.get("phase_ms") or r.get("stage_ms"), "fp32_accurate", j.get("fp32_accurate"))
    for k, v in (j.get("secondary") or {}).items():
        if isinstance(v, dict):
            print("   ", k, {kk: (float("%.4g" % vv) if isinstance(vv, float) else vv) for kk, vv in v.items() if kk in
                             ("ms_per_step", "parity_max_abs_err", "layer_frac", "gpu_ms_per_step", "emulated_copy_ms", "gpu_ms_per_step_minus_emulated_copy", "stage_ms", "halo_rows", "link_ms_at_70pct_of_7x76.8GBs", "phase_ms")})
PY
}
case $MODE in
quick)
    # (separate processes: a faulting kernel poisons every later test of its process)
    grp() { timeout 900 python -m pytest tests -m gpu -q -k "$2" > gpurun_out/pytest_$1.log 2>&1; grep -E "^FAILED|^ERROR| passed| failed" gpurun_out/pytest_$1.log | tail -12; grep -E "^E  " gpurun_out/pytest_$1.log | head -12; }
    grp A "target_block or real_halos or 24_bit or two_rank"
    grp B "bucketed or staged or prepared or strict or in_place or reference_call or deterministic or f16_split_rows or partitioned_graph_on_gpu"
    grp C "matches_oracle or fused_aggregate or golden"
    for n in 768 512; do python tools/bench_linear.py --which bf16x3 --n-out $n --c24 2>&1 | grep -v "^ *trace"; done > gpurun_out/linear_$TAG.log 2>&1
    cat gpurun_out/linear_$TAG.log
    for loc in 0 0.5 0.75 0.9; do
        timeout 300 python bench.py --emulate-world 8 --locality $loc --steps 5 > gpurun_out/emu8_loc$loc.json 2> gpurun_out/emu8_loc$loc.err || tail -3 gpurun_out/emu8_loc$loc.err
    done
    timeout 300 python bench.py --emulate-world 8 --blocks 16 --steps 5 > gpurun_out/emu8_b16.json 2> gpurun_out/emu8_b16.err
    timeout 300 python bench.py --emulate-world 8 --halo-fp32 --steps 5 > gpurun_out/emu8_fp32.json 2> gpurun_out/emu8_fp32.err
    summ gpurun_out/emu8_*.json
    for p in bf16x3 f16x3; do timeout 300 python bench.py --precision $p --no-secondary --no-cpu-baseline > gpurun_out/bench_quick_$p.json 2> gpurun_out/bench_quick_$p.err; done
    summ gpurun_out/bench_quick_*.json
    ;;
validate)
    python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/pytest_$TAG.log
    tail -8 gpurun_out/pytest_$TAG.log
    ( time timeout 900 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err ) 2>&1 | grep real
    HGT_BENCH_DEVICE=0 HGT_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 --nodes-per-gpu 200000 --edges-per-gpu 2000000 --no-cpu-baseline > gpurun_out/bench_${TAG}_dist2.json 2> gpurun_out/bench_${TAG}_dist2.err
    summ gpurun_out/bench_$TAG.json gpurun_out/bench_${TAG}_dist2.json
    tail -c 400 gpurun_out/bench_${TAG}_dist2.err
    python tools/fuzz_parity.py 60 > gpurun_out/fuzz_$TAG.log 2>&1; tail -1 gpurun_out/fuzz_$TAG.log; grep -c FAIL gpurun_out/fuzz_$TAG.log
    python tools/bench_small.py > gpurun_out/bench_small_$TAG.log 2>&1; tail -2 gpurun_out/bench_small_$TAG.log
    python tools/bench_train.py > gpurun_out/bench_train_$TAG.log 2>&1; tail -2 gpurun_out/bench_train_$TAG.log
    ;;
profile)
    tools/profile_pmc.sh $TAG > gpurun_out/prof_$TAG.log 2>&1
    python - <<PY
import json
j = json.load(open("gpurun_out/prof_$TAG/${TAG}_pmc_summary.json"))
for k in ("avg_kernel_ms", "traffic_bytes", "mfma_busy_pct", "valu_busy_pct", "waves_per_simd_avg", "wait_pct", "lds_bank_conflict_pct"): print(k, j.get(k))
PY
    $0 latency $TAG f16x3
    ;;
final)
    timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
    timeout 300 python examples/train_synthetic.py --steps 6 2>&1 | tail -3
    python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/pytest_$TAG.log
    grep -E "^FAILED|^ERROR| passed| failed" gpurun_out/pytest_$TAG.log | tail -10
    # the x-stationary GEMM (1024) and the fused sub-tile aggregation (64) forced onto every layer of the backward / staged / oracle tests
    # (their size thresholds keep them off small graphs otherwise).  The LDS-ring aggregation / single-pass runs (512 / 256) exist in LAB
    # builds only (make LAB=1: tools/lab/build_lab.sh compiles them; the shipped library ignores the bits)
    HGT_TEST_KERNEL_FLAGS=$((1024 + 64)) timeout 900 python -m pytest tests -m gpu -q -k "backward or staged or two_rank or matches_oracle or fused or golden" 2>&1 | tail -3
    timeout 300 python tools/bench_xs.py > gpurun_out/${TAG}_xs_check.log 2>&1; grep -v "BIT-IDENTICAL (" gpurun_out/${TAG}_xs_check.log | tail -12
    tools/profile_pmc.sh $TAG > gpurun_out/prof_$TAG.log 2>&1
    cp gpurun_out/prof_$TAG/${TAG}_pmc_summary.json profiles/${TAG}_pmc_summary.json      # the line below is judged against THIS build's counters
    # per-kernel timelines of the sampled-batch workloads at this build, both split precisions (round-5 review, task 7b)
    $0 latency $TAG f16x3 > gpurun_out/latency_$TAG.log 2>&1; $0 latency $TAG bf16x3 >> gpurun_out/latency_$TAG.log 2>&1
    cp gpurun_out/${TAG}_latency_*.txt profiles/ 2>/dev/null
    tools/profile_train.sh $TAG > gpurun_out/prof_train_$TAG.log 2>&1
    python tools/bench_train.py 2>/dev/null | tail -1 > gpurun_out/${TAG}_train_line.json
    for loc in 0 0.5 0.75 0.9; do
        timeout 300 python bench.py --emulate-world 8 --locality $loc --steps 7 > gpurun_out/${TAG}_emu8_loc$loc.json 2> gpurun_out/${TAG}_emu8_loc$loc.err
    done
    summ gpurun_out/${TAG}_emu8_loc*.json
    ( time timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err ) 2>&1 | grep real
    summ gpurun_out/${TAG}_bench.json
    ;;
xs)
    timeout 420 python tools/bench_xs.py > gpurun_out/xs_check_$TAG.log 2>&1; echo "bench_xs rc=$?"
    grep -v "BIT-IDENTICAL (" gpurun_out/xs_check_$TAG.log | tail -30
    for m in 0 1; do
        timeout 300 python bench.py --kernel-flags $((m ? 1024 : 2048)) --no-secondary --no-cpu-baseline --steps 40 > gpurun_out/xs_c2_$m.json 2> gpurun_out/xs_c2_$m.err
        timeout 300 python bench.py --kernel-flags $((m ? 1024 : 2048)) --dim 512 --nodes-per-gpu 500000 --edges-per-gpu 5000000 --no-secondary --no-cpu-baseline > gpurun_out/xs_d512_$m.json 2> gpurun_out/xs_d512_$m.err
    done
    summ gpurun_out/xs_c2_0.json gpurun_out/xs_c2_1.json gpurun_out/xs_d512_0.json gpurun_out/xs_d512_1.json
    ;;
judged)
    # the judged command under rocprofv3 (kernel statistics + counter passes), then the judged line against this build's counters
    tools/profile_pmc.sh $TAG > gpurun_out/prof_$TAG.log 2>&1
    cp gpurun_out/prof_$TAG/${TAG}_pmc_summary.json profiles/${TAG}_pmc_summary.json
    ( time timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err ) 2>&1 | grep real
    summ gpurun_out/${TAG}_bench.json
    ;;
suite)
    HGT_TEST_KERNEL_FLAGS=${3:-0} timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -8 | tee gpurun_out/suite_${TAG}_flags${3:-0}.txt
    ;;
sanity)
    timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
    timeout 300 python bench.py --no-secondary --no-cpu-baseline > gpurun_out/sanity_bench.json 2> gpurun_out/sanity_bench.err; echo "bench rc=$?"
    summ gpurun_out/sanity_bench.json
    timeout 300 python -m pytest tests -m gpu -q -x -k "golden or xs_gemm or reference_call or ring_aggregation" 2>&1 | tail -2
    ;;
ceiling)
    timeout 200 python tools/lab/stream_ceiling.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/stream_ceiling.txt
    ;;
xsthr)
    timeout 300 python tools/bench_xs.py --threshold 2>&1 | grep -v amdgpu.ids | tee gpurun_out/xs_threshold.txt
    ;;
latency)
    export TMPDIR=/tmp; ROOT=$(pwd); PREC=${3:-f16x3}; cd /tmp
    for wl in c3:1 c3w520:1 c5:2 mag4:4 c1:1; do
        w=${wl%%:*}; nl=${wl##*:}
        rm -rf /tmp/pl_$w
        timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/pl_$w -o t -- python $ROOT/tools/trace_latency.py run $w $PREC > /tmp/pl_$w.log 2>&1
        { echo "# tools/trace_latency.py $w $PREC  ($(grep -E '^N ' /tmp/pl_$w.log | tail -1))  commit $HGT_COMMIT"; python $ROOT/tools/trace_latency.py show /tmp/pl_$w $nl; } > $ROOT/gpurun_out/${TAG}_latency_${w}_$PREC.txt 2>&1
        tail -1 $ROOT/gpurun_out/${TAG}_latency_${w}_$PREC.txt
    done
    cd $ROOT
    ;;
small)
    timeout 600 python bench.py --small-only > gpurun_out/${TAG}_small.json 2> gpurun_out/${TAG}_small.err; echo "rc=$?"; tail -3 gpurun_out/${TAG}_small.err
    python - gpurun_out/${TAG}_small.json <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
for k, v in j.items():
    if not isinstance(v, dict): continue
    print(k, {kk: vv for kk, vv in v.items() if kk != "workload" and not isinstance(vv, dict)})
    for p, e in v.items():
        if isinstance(e, dict): print("    ", p, {kk: (float("%.4g" % vv) if isinstance(vv, float) else vv) for kk, vv in e.items()})
PY
    ;;
emuprof)
    export TMPDIR=/tmp; ROOT=$(pwd); cd /tmp
    for args in "--locality 0" "--locality 0.75"; do
        rm -rf /tmp/pe
        timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pe -o s -- python $ROOT/bench.py --emulate-world 8 $args --steps 5 > /tmp/pe.log 2>&1
        python - "$args" <<'PY'
import csv, glob, sys
f = glob.glob("/tmp/pe/**/*kernel_stats.csv", recursive=True)
rows = sorted(csv.DictReader(open(f[0])), key=lambda r: -float(r["TotalDurationNs"]))
print("bench.py --emulate-world 8", sys.argv[1])
for r in rows[:10]:
    print("%-90s %6s %10.1f us %10.2f ms" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
    done
    ;;
esac
