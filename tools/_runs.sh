cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export HGT_LIB_PATH=$GRAFT_REPO_ROOT/pyhgt_amd/lib_dev/libhgt_hip.so
python -m pytest tests/test_hgt_gpu.py -m gpu -q -x -k "bucketed or pipelined or source_only or partitioned" 2>&1 | tail -30
