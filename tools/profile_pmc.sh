#!/bin/bash
# rocprofv3 passes of the judged benchmark command (run on the GPU box through gpurun, from the repo root):
#   tools/profile_pmc.sh <tag> [extra bench.py flags]
# writes gpurun_out/prof_<tag>/{stats,fetch,write,sqA..sqE}/...; tools/pmc_summary.py turns them into profiles/<tag>_*.csv|json.
# Counters are collected in their own runs (kernel trace only next to --pmc), one small set per pass.
set -u
TAG=${1:-r03}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-secondary $*"
cd /tmp
rocprofv3 -L 2>&1 | grep -oE "^\s*(gpu-agent[0-9]+:)?\s*[A-Za-z_0-9]+" | sort -u | head -400 > "$OUT/counters_available.txt" || true
# the timing pass runs more steps than the counter passes: with 4 launches the average is dominated by the first (cold) one
BENCH_T="python $ROOT/bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-parity --no-secondary $*"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o s -- $BENCH_T > "$OUT/stats.log" 2>&1
pass() {  # name counters...
    local name=$1; shift
    timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d "$OUT/$name" -o p -- $BENCH > "$OUT/$name.log" 2>&1 || echo "pass $name failed" >> "$OUT/failed.txt"
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sqA SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES
pass sqB SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS
pass sqB2 SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_BF16
pass sqC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pass sqD SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS
pass sqE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM
pass tcc TCC_HIT_sum TCC_MISS_sum
COMMIT=${HGT_COMMIT:-unknown}
for l in "$OUT"/*.log; do tail -c 1500 "$l" > "$l.tail"; rm -f "$l"; done
python "$ROOT/tools/pmc_summary.py" "$OUT" "$TAG" "$COMMIT"
du -sh "$OUT"; ls "$OUT"
