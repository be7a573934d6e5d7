cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; ulimit -c 0; export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
cd /tmp
for args in "--locality 0" "--locality 0 --blocks 5 --block-shape geometric"; do
  rm -rf /tmp/pe
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pe -o s -- python $ROOT/bench.py --emulate-world 8 $args --steps 5 > /tmp/pe.log 2>&1
  tail -1 /tmp/pe.log | cut -c1-500
  python - "$args" <<'PY'
import csv, glob, sys
f = glob.glob("/tmp/pe/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
print("args", sys.argv[1])
for r in rows[:10]:
    print("%-90s %6s %10.1f %10.2f" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
done
