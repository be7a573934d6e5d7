#!/bin/bash
# tools/lab/pmc_store_path.sh: vector-memory / store-path counters (TA, TCP, TCC write side) of the three kernels of the judged layer,
# per launch -- the counter-backed view of what the Q|K|V kernel's stores cost (round-4 review, item 7).  Separate --pmc passes, kernel
# trace only.  Writes gpurun_out/r05_store_path.txt
cd ${GRAFT_REPO_ROOT:-$(pwd)}; ROOT=$(pwd); mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
pass() { local name=$1; shift; rm -rf /tmp/sp_$name; timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d /tmp/sp_$name -o p -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-secondary > /tmp/sp_$name.log 2>&1 || { echo "pass $name failed"; tail -3 /tmp/sp_$name.log; }; }
pass a GRBM_GUI_ACTIVE TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum
pass b TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum
pass c TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum SQ_INST_CYCLES_VMEM_WR
pass d TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCC_BUSY_avr SQ_INSTS_VMEM_WR
python - <<'PY' | tee $ROOT/gpurun_out/r05_store_path.txt
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
for path in glob.glob("/tmp/sp_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"]
        lab = "project_qkv" if "k_typed_linear_xs" in k else "edge_logits" if "k_edge_logits" in k else "edge_aggregate" if "k_edge_aggregate_update" in k else None
        if lab is None: continue
        acc[lab][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[(lab, row["Counter_Name"])].add(row["Dispatch_Id"])
names = sorted({c for d in acc.values() for c in d})
print("%-36s %16s %16s %16s" % ("counter (per launch)", "project_qkv", "edge_logits", "edge_aggregate"))
for c in names:
    print("%-36s %16.5g %16.5g %16.5g" % tuple([c] + [acc[l][c] / max(1, len(cnt[(l, c)])) for l in ("project_qkv", "edge_logits", "edge_aggregate")]))
PY
