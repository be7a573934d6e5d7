cd $GRAFT_REPO_ROOT
for extra in "" "--rte"; do
python bench.py $extra --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[$extra]', 'ms', round(j['ms_per_step'],3), 'parity', j['parity_max_abs_err'], j['roofline']['phase_ms'])
"
done
