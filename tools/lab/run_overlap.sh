cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; ulimit -c 0
timeout 900 python -m pytest tests -m gpu -q -k "target_block or real_halos or deterministic or two_rank or hub" 2>&1 | tail -4
for loc in 0 0.75; do timeout 300 python bench.py --emulate-world 8 --locality $loc --steps 5 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('loc', j['locality'], 'gpu_ms', j['gpu_ms_per_step'], j['stage_ms'])"; done
