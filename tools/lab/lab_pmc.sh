#!/bin/bash
# tools/lab: rocprofv3 counter passes of tools/bench_linear.py for a list of wide-kernel variants (and the round-2 kernel):
#   tools/lab_pmc.sh 0 1 10 pc
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/lab_pmc
export TMPDIR=/tmp
export HGT_LIB_PATH=$GRAFT_REPO_ROOT/pyhgt_amd/lib_lab/libhgt_hip.so
ROOT=$GRAFT_REPO_ROOT
cd /tmp
for v in "$@"; do
  EXTRA=""; if [ "$v" = "pc" ]; then unset HGT_WD_VARIANT; else export HGT_WD_VARIANT=$v; fi
  for pass in A B; do
    if [ $pass = A ]; then C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU";
    else C="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; fi
    rm -rf /tmp/lp_$v$pass
    timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc $C -d /tmp/lp_$v$pass -o p -- python $ROOT/tools/bench_linear.py --which bf16x3 --iters 3 $EXTRA > /tmp/lp_$v$pass.log 2>&1
  done
  python - "$v" <<'PY' | tee -a $ROOT/gpurun_out/lab_pmc/summary.txt
import csv, glob, sys, collections
v = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for p in "AB":
    for path in glob.glob("/tmp/lp_%s%s/**/*counter_collection.csv" % (v, p), recursive=True):
        for row in csv.DictReader(open(path)):
            k = row["Kernel_Name"]
            if "typed_linear" not in k: continue
            acc[k[:60]][row["Counter_Name"]] += float(row["Counter_Value"]); n[(k[:60], row["Counter_Name"])].add(row["Dispatch_Id"])
for k, c in acc.items():
    c = {m: val / max(1, len(n[(k, m)])) for m, val in c.items()}
    cyc = c.get("GRBM_GUI_ACTIVE", 0) / 8
    out = {"variant": v, "kernel": k, "cycles": round(cyc), "mfma_busy_pct": round(100 * c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (cyc * 1024 + 1e-9), 1),
           "waves_per_simd": round(4 * c.get("SQ_WAVE_CYCLES", 0) / (cyc * 1024 + 1e-9), 2),
           "wait_any_pct": round(100 * c.get("SQ_WAIT_ANY", 0) / (c.get("SQ_WAVE_CYCLES", 1)), 1),
           "wait_inst_pct": round(100 * c.get("SQ_WAIT_INST_ANY", 0) / (c.get("SQ_WAVE_CYCLES", 1)), 1),
           "active_pct": round(100 * c.get("SQ_ACTIVE_INST_ANY", 0) / (c.get("SQ_WAVE_CYCLES", 1)), 1),
           "valu_busy_pct": round(100 * 4 * c.get("SQ_ACTIVE_INST_VALU", 0) / (cyc * 1024 + 1e-9), 1),
           "insts_valu": round(c.get("SQ_INSTS_VALU", 0)), "insts_salu": round(c.get("SQ_INSTS_SALU", 0)),
           "vmem_rd": round(c.get("SQ_INSTS_VMEM_RD", 0)), "vmem_wr": round(c.get("SQ_INSTS_VMEM_WR", 0)),
           "lds_active": round(c.get("SQ_ACTIVE_INST_LDS", 0)), "lds_wait_inst": round(c.get("SQ_WAIT_INST_LDS", 0)),
           "lds_conflict": round(c.get("SQ_LDS_BANK_CONFLICT", 0)), "lds_idx_active": round(c.get("SQ_LDS_IDX_ACTIVE", 0))}
    print(out)
PY
done
tail -3 /tmp/lp_*A.log | tail -20
