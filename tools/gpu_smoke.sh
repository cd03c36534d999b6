cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_full_size.py -m gpu -x -q -k closed_loop 2>&1 | tail -12
