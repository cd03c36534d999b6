cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -m gpu -x -q 2>&1 | tail -3
python tools/gpu_solve_cycles.py 4096 2>&1 | tail -7
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-hji 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ['value','ms_per_step','phase_ms','warm_value','solved','ipm_iters_mean']})"
