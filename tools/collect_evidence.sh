#!/bin/bash
# tools/collect_evidence.sh <tag>: copy what `tools/gpu.sh final <tag>` left in gpurun_out/ (scratch, merged back from the GPU box) into
# profiles/ (tracked): the judged line, rocprofv3 kernel statistics + counters, latency timelines, training line, rank-of-8 emulation,
# x-stationary GEMM check, test-suite summaries.  Run from the repo root in the build container after the gpurun call has returned.
set -e
TAG=${1:-r06}; G=gpurun_out; P=profiles
COMMIT=$(cat .commit 2>/dev/null || echo unknown)
cp $G/prof_$TAG/${TAG}_kernel_stats.csv $G/prof_$TAG/${TAG}_pmc_per_kernel.csv $G/prof_$TAG/${TAG}_pmc_summary.json $P/
tail -1 $G/${TAG}_bench.json | python -c "import json,sys; print(json.dumps(json.loads(sys.stdin.read()), indent=1))" > $P/${TAG}_bench_line.json
cp $G/${TAG}_latency_*.txt $P/
cp $G/${TAG}_train_line.json $G/${TAG}_train_kernel_stats.txt $P/ 2>/dev/null || true
python - $TAG <<'PY'
import json, sys, glob, os
tag = sys.argv[1]
out = {}
for f in sorted(glob.glob("gpurun_out/%s_emu8_loc*.json" % tag)):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        out[os.path.basename(f)] = {"unreadable": str(e)}; continue
    out[os.path.basename(f)] = {k: j.get(k) for k in ("locality", "halo_rows", "gpu_ms_per_step", "emulated_copy_ms", "gpu_ms_per_step_minus_emulated_copy",
                                                     "gpu_ms_per_step_all", "stage_ms", "link_ms_at_70pct_of_7x76.8GBs")}
json.dump(out, open("profiles/%s_rank_of_8_emulation.json" % tag, "w"), indent=1)
PY
grep -E "BIT-IDENTICAL$|^ALL|^time|DIFFER" $G/${TAG}_xs_check.log | tail -14 > $P/${TAG}_xs_gemm_check.txt
grep -E " passed| failed" $G/pytest_$TAG.log | tail -1 > $P/${TAG}_gpu_suite.txt
{ echo "tools/gpu.sh final $TAG (commit $COMMIT), forced-kernel pass:"
  echo '  HGT_TEST_KERNEL_FLAGS=$((1024 + 64)) pytest -m gpu -k "backward or staged or two_rank or matches_oracle or fused or golden"'
  echo "  (1024 = x-stationary GEMM on every eligible typed linear, 64 = fused sub-tile aggregation + update at every size)"
  grep -E "passed.*skipped.*deselected" $G/final_$TAG.log | sed -n 2p | sed 's/^/  -> /'
  echo "  (the tests that compare two item-parallel forms skip themselves when flag 64 is forced: neither form runs then)"; } > $P/${TAG}_forced_kernels_suite.txt
echo "profiles/${TAG}_* refreshed at $COMMIT:"; python - $TAG <<'PY'
import json, sys
j = json.load(open("profiles/%s_bench_line.json" % sys.argv[1])); r = j["roofline"]
print(" ms", round(j["ms_per_step"], 3), "layer_frac", r["layer_frac"], "frac", r["frac"], "pmc_commit", r.get("pmc_commit"), "pmc_stale", r.get("pmc_stale"),
      "c3", round(j["c3_us_per_layer"], 1), "c5", round(j["c5_us_per_forward"], 1), "mag4", round(j["mag4_us_per_layer"], 1))
PY
