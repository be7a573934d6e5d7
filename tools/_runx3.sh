cd $GRAFT_REPO_ROOT
export HGT_LIB_PATH=$GRAFT_REPO_ROOT/pyhgt_amd/lib_sp/libhgt_hip.so
for fs in 1 2 4; do
export HGT_FORCE_SPLIT=$fs
for extra in "--kernel-flags 1" "--kernel-flags 1 --rte"; do
python bench.py $extra --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('split $fs [$extra]', 'ms', round(j['ms_per_step'],3), 'parity', j['parity_max_abs_err'], j['roofline']['phase_ms'])
"
done
done
