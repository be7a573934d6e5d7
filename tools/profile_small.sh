#!/bin/bash
# rocprofv3 kernel statistics of the latency-regime workloads (tools/bench_small.py) -> gpurun_out/<tag>_small_kernel_stats.txt
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=${1:-r03}
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
cd /tmp
rm -rf /tmp/ps
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps -o s -- python $ROOT/tools/bench_small.py > /tmp/ps.log 2>&1
tail -4 /tmp/ps.log
python - <<'PY' > $ROOT/gpurun_out/${TAG}_small_kernel_stats.txt
import csv, glob
f = glob.glob("/tmp/ps/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
print("%-100s %8s %10s %9s" % ("kernel", "calls", "avg_us", "pct"))
for r in rows[:40]:
    print("%-100s %8s %10.2f %9s" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
head -45 $ROOT/gpurun_out/${TAG}_small_kernel_stats.txt
