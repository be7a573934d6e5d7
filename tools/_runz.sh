cd $GRAFT_REPO_ROOT
python -m pytest tests/test_backward_gpu.py -m gpu -q -x 2>&1 | tail -4
python tools/bench_train.py
HGT_TRAIN_N=3200 HGT_TRAIN_E=31000 python tools/bench_train.py
