#!/bin/bash
# tools/lab/pmc_quick.sh <libname> : wave-cycle / wait / instruction counters of the aggregation kernel of the judged layer on a lab library
cd ${GRAFT_REPO_ROOT:-$(pwd)}; ROOT=$(pwd); mkdir -p gpurun_out
n=$1
case $n in prod) L=pyhgt_amd/lib/libhgt_hip.so;; dev) L=pyhgt_amd/lib_lab/libhgt_hip.so;; *) L=pyhgt_amd/lib_lab_$n/libhgt_hip.so;; esac
export HGT_LIB_PATH=$ROOT/$L TMPDIR=/tmp
cd /tmp
pass() { local name=$1; shift; rm -rf /tmp/pq_$name; timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d /tmp/pq_$name -o p -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-secondary > /tmp/pq_$name.log 2>&1 || echo "pass $name failed"; }
pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS
pass b SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU
pass c SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC
python - $n <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
for path in glob.glob("/tmp/pq_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"]
        if "aggregate" not in k: continue
        k = k.replace("(anonymous namespace)::", "")[:60]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[(k, row["Counter_Name"])].add(row["Dispatch_Id"])
for k, d in acc.items():
    print(sys.argv[1], k)
    for c, v in sorted(d.items()):
        print("   %-24s %.4g per launch" % (c, v / max(1, len(cnt[(k, c)]))))
PY
