cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; ulimit -c 0
timeout 900 python -m pytest tests -m gpu -q -k "item_parallel or golden or matches_oracle or dense_hgt or prepared or gnn_wrapper" 2>&1 | tail -15
python tools/bench_small.py 2>&1 | tail -1
