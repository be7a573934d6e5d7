cd $GRAFT_REPO_ROOT
python -m pytest tests/test_backward_gpu.py -m gpu -q -x 2>&1 | tail -15
