cd $GRAFT_REPO_ROOT
for lib in $HGT_LIBS; do
export HGT_LIB_PATH=$GRAFT_REPO_ROOT/pyhgt_amd/$lib/libhgt_hip.so
for extra in "" "--kernel-flags 1" $HGT_EXTRA; do
python bench.py $extra --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib [$extra]', 'ms', round(j['ms_per_step'],3), 'parity', j['parity_max_abs_err'], j['roofline']['phase_ms'])
"
done
done
