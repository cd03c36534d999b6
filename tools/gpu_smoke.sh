cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_hji.py -m gpu -x -q 2>&1 | tail -3
python bench.py --steps 10 --warmup 2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ['value','ms_per_step','hji_lookup','cpu_baseline']})"
