#!/bin/bash
# tools/lab/pmc_gemm.sh "<rows k n_out>" <kernel-name substring>: wave-cycle / wait / matrix-core counters of a latency-regime typed linear
# (tools/lab/tile_gemm.py: flush kernel + 3 launches of the tile kernel + 3 of the slab kernel per iteration)
cd ${GRAFT_REPO_ROOT:-$(pwd)}; ROOT=$(pwd); mkdir -p gpurun_out; export TMPDIR=/tmp; cd /tmp
shape=$1; pat=${2:-k_tile_linear}
pass() { local name=$1; shift; rm -rf /tmp/pg_$name; timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d /tmp/pg_$name -o p -- python $ROOT/tools/lab/tile_gemm.py run $shape > /tmp/pg_$name.log 2>&1 || echo "pass $name failed"; }
pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES
pass b SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD
pass c SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU
pass d TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum
pass e SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS
python - "$pat" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
for path in glob.glob("/tmp/pg_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"]
        if sys.argv[1] not in k and "typed_linear" not in k: continue
        k = k.replace("(anonymous namespace)::", "")[:70]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[(k, row["Counter_Name"])].add(row["Dispatch_Id"])
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-32s %.5g per launch" % (c, v / max(1, len(cnt[(k, c)]))))
PY
