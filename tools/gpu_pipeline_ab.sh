# A/B of the pipelined nodes + update_QP launch (PG_PIPELINE=0 / 1) on config 3 (fp32 + HJI row), after the bit-identity tests
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "pipelined" > gpurun_out/pytest_pipe.log 2>&1; tail -3 gpurun_out/pytest_pipe.log
B="bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-decoupled --no-rollout --no-warm"
for m in 0 1; do
  PG_PIPELINE=$m timeout -k 10 300 python $B > gpurun_out/bench_c3_pipe$m.log 2>&1
  tail -1 gpurun_out/bench_c3_pipe$m.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); f=d['fp32']; print('PG_PIPELINE=$m', round(d['value']), 'config3', round(f['value']), f['ms_per_step'], f['phase_ms'], f['solved'], 'without_hji', round(f['without_hji']))"
done
timeout -k 10 300 python bench.py --precision f32 --batch 8192 --no-hji --no-decoupled > gpurun_out/bench_f32_8192.log 2>/dev/null; tail -1 gpurun_out/bench_f32_8192.log | cut -c1-400
