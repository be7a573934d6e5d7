#!/bin/bash
# tools/lab/pcsamp.sh <libname> [method=host_trap] [unit=time] [interval]: PC sampling of the judged layer (beta feature of rocprofv3)
cd ${GRAFT_REPO_ROOT:-$(pwd)}; ROOT=$(pwd); mkdir -p gpurun_out
n=$1; M=${2:-host_trap}; U=${3:-time}; I=${4:-1}
case $n in prod) L=pyhgt_amd/lib/libhgt_hip.so;; dev) L=pyhgt_amd/lib_lab/libhgt_hip.so;; *) L=pyhgt_amd/lib_lab_$n/libhgt_hip.so;; esac
export HGT_LIB_PATH=$ROOT/$L TMPDIR=/tmp
cd /tmp; rm -rf /tmp/pcs
timeout 240 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit $U --pc-sampling-method $M --pc-sampling-interval $I --kernel-trace --output-format csv -d /tmp/pcs -o p -- python $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-parity --no-secondary > /tmp/pcs.log 2>&1
echo "rc=$?"; tail -5 /tmp/pcs.log; find /tmp/pcs -type f | head; 
python - <<'PY'
import csv, glob, collections
fs = glob.glob("/tmp/pcs/**/*pc_sampling*.csv", recursive=True)
print(fs)
for f in fs[:2]:
    rows = list(csv.DictReader(open(f)))
    print(f, len(rows), rows[0].keys() if rows else None)
    c = collections.Counter()
    for r in rows:
        c[(r.get("Instruction") or r.get("Instruction_Comment") or "?")[:70]] += 1
    for k, v in c.most_common(60):
        print("%6d  %s" % (v, k))
PY
cp /tmp/pcs.log $ROOT/gpurun_out/pcs_$n.log 2>/dev/null
