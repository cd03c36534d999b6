# A/B of the pipelined nodes + update_QP launch: PG_PIPE_PUB = hex mask of the nodes after which the recurrence publishes (every publication is a device-scope release)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B="bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-hji --no-decoupled --no-f32 --no-rollout --no-warm"
show() { tail -1 $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2', round(d['value']), round(d['ms_per_step'],4), [round(v,4) for v in d['phase_ms'].values()], d['solved'])"; }
for m in ffffffffffffffff 92493fe 12492554 1084248 80200 0; do
  PG_PIPE_PUB=$m timeout -k 10 200 python $B > gpurun_out/bench_m$m.log 2>&1; show gpurun_out/bench_m$m.log "f64 4096 pub=$m"
  PG_PIPE_PUB=$m timeout -k 10 200 python $B --batch 16384 > gpurun_out/bench_16384_m$m.log 2>&1; show gpurun_out/bench_16384_m$m.log "f64 16384 pub=$m"
  PG_PIPE_PUB=$m timeout -k 10 200 python $B --precision f32 --batch 8192 > gpurun_out/bench_f32_m$m.log 2>&1; show gpurun_out/bench_f32_m$m.log "f32 8192 pub=$m"
done
