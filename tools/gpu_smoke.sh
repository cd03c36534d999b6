cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-hji --no-decoupled --no-f32 > gpurun_out/bench.log 2>&1
tail -1 gpurun_out/bench.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ['value','ms_per_step','phase_ms','warm_value','solved','ipm_iters_mean']})"
