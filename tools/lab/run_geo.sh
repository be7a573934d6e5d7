cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; ulimit -c 0
for args in "--locality 0" "--locality 0 --block-shape geometric" "--locality 0 --blocks 5 --block-shape geometric" "--locality 0.75 --block-shape geometric" "--locality 0.75 --blocks 4"; do
timeout 300 python bench.py --emulate-world 8 $args --steps 5 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$args', 'gpu_ms', j['gpu_ms_per_step'], j['stage_ms'], j['halo_rows_per_chunk'])"; done
