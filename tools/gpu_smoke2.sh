cd $GRAFT_REPO_ROOT
python tools/gpu_sweep.py 2>&1 | grep mu0
