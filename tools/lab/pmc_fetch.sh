#!/bin/bash
# one counter pass (FETCH_SIZE + TCC hit/miss) of the judged command: HBM read bytes per launch of the hot kernels
#   tools/lab/pmc_fetch.sh <tag> [bench flags]
TAG=${1:-x}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/fetch_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $c -d /tmp/pf_$n -o p -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-secondary "$@" > /tmp/pf_$n.log 2>&1
done
python - <<'PY' > $OUT/summary.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/pf_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        for key in ("k_edge_logits", "k_edge_aggregate", "k_typed_linear_pc"):
            if key in k:
                acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    for cn, vals in v.items():
        vals = vals[1:] if len(vals) > 1 else vals      # drop the cold launch
        m = sum(vals) / len(vals)
        extra = " -> %.2f GB (x2 x1024)" % (2 * m * 1024 / 1e9) if cn == "FETCH_SIZE" else ""
        print("%-20s %-14s avg %.4g over %d launches%s" % (k, cn, m, len(vals), extra))
PY
cat $OUT/summary.txt
