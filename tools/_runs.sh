cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export HGT_LIB_PATH=$GRAFT_REPO_ROOT/pyhgt_amd/lib_dev/libhgt_hip.so
python -m pytest tests/test_hgt_gpu.py -m gpu -q -x -k "sorted or malformed or sampled" 2>&1 | tail -5
python tools/bench_small.py 2>&1 | grep "c3 surrogate"
