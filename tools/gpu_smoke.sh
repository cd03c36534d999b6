cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5
python bench.py --steps 10 --warmup 2 2>&1 | tail -5
