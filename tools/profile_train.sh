#!/bin/bash
# rocprofv3 kernel statistics of one training step at c2 (tools/bench_train.py) -> gpurun_out/<tag>_train_kernel_stats.txt
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=${1:-r03}
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
cd /tmp
rm -rf /tmp/pt
HGT_TRAIN_NO_SMALL=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt -o s -- python $ROOT/tools/bench_train.py > /tmp/pt.log 2>&1
tail -2 /tmp/pt.log
python - <<'PY' > $ROOT/gpurun_out/${TAG}_train_kernel_stats.txt
import csv, glob
f = glob.glob("/tmp/pt/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
print("%-110s %8s %10s %10s %8s" % ("kernel", "calls", "avg_us", "total_ms", "pct"))
for r in rows[:45]:
    print("%-110s %8s %10.1f %10.2f %8s" % (r["Name"][:110], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
head -50 $ROOT/gpurun_out/${TAG}_train_kernel_stats.txt
