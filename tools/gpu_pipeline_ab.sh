# The pipelined nodes + update_QP launch against launch-per-phase on config 3 (fp32 + HJI row), after the bit-identity tests.  (Since round 5 the library reads no
# environment: the bench measures both forms itself -- "launch_per_phase" in bench_full.json.)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "pipelined" > gpurun_out/pytest_pipe.log 2>&1; tail -3 gpurun_out/pytest_pipe.log
B="bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-decoupled --no-rollout --no-warm"
for m in 1; do
  timeout -k 10 300 python $B --full-record gpurun_out/bench_c3_pipe$m.json > gpurun_out/bench_c3_pipe$m.log 2>&1
  tail -1 gpurun_out/bench_c3_pipe$m.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); f=d['fp32']; print('PG_PIPELINE=$m', round(d['value']), 'config3', round(f['value']), f['ms_per_step'], f['phase_ms'], f['solved'], 'without_hji', round(f['without_hji']))"
done
timeout -k 10 300 python bench.py --precision f32 --batch 8192 --no-hji --no-decoupled > gpurun_out/bench_f32_8192.log 2>/dev/null; tail -1 gpurun_out/bench_f32_8192.log | cut -c1-400
