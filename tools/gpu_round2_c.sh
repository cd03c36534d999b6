cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py > gpurun_out/bench_full.log 2> gpurun_out/bench_full.err; tail -c 1500 gpurun_out/bench_full.log; tail -3 gpurun_out/bench_full.err
python bench.py --precision f32 --batch 8192 --no-hji --no-decoupled > gpurun_out/bench_f32_8192.log 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/bench_f32_8192.log').read().strip().splitlines()[-1]); print({k:d[k] for k in ['value','ms_per_step','phase_ms','dtype']}, d['cpu_baseline']['accuracy'])"
python bench.py --gpus 2 --backend gloo --steps 5 --warmup 1 --no-cpu-baseline --no-hji --no-decoupled --no-f32 2>/dev/null | tail -1 | cut -c1-300
bash tools/gpu_profile.sh
