cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-hji --no-decoupled > gpurun_out/bench2.log 2>&1
tail -1 gpurun_out/bench2.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ['value','ms_per_step','phase_ms','warm_value','solved','ipm_iters_mean']}); print(d.get('fp32'))"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_b -o run -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-hji --no-decoupled --no-f32 > gpurun_out/prof_b.log 2>&1
find gpurun_out/prof_b -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'cut -d, -f1-4 {} | cut -c1-150 | head -12'
