# A/B of alternative builds of the fp64 library (pigeon.jl_amd/csrc/libpigeon_hip_rg*.so) against the shipped one: cold headline batch (rounds, interior-point
# instances, phase times) and closed loop; PG_AB_FULL=1 adds the all-paths probe and the solver tests
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B="bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-hji --no-decoupled --no-f32"
run() {
  timeout -k 10 300 python $B > gpurun_out/bench_$1.log 2>&1
  tail -1 gpurun_out/bench_$1.log | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['closed_loop_rollout']['warm_start_of_active_set']
print('$1', round(d['value']), [round(x,4) for x in d['phase_ms'].values()], d['solved'], 'ipm', d['ipm_iters_hist'], 'rounds', d['polish_rounds_hist'], 'warm', round(d['warm_value']), 'closed loop', round(r['value']), r['served_by_warm_polish_alone'])"
  if [ "$PG_AB_FULL" = "1" ]; then timeout -k 10 300 python tools/gpu_all_paths_probe.py 2>&1 | grep -v amdgpu.ids | tail -9; fi
}
run base
# (the alternative build is SELECTED through PIGEON_HIP_LIB -- pigeon.jl_amd/_lib.py --, never copied over the shipped library: a timeout in between used to leave it there)
for v in pigeon.jl_amd/csrc/libpigeon_hip_rg*.so; do
  export PIGEON_HIP_LIB=$PWD/$v
  run $(basename $v .so)
  if [ "$PG_AB_FULL" = "1" ]; then timeout -k 10 400 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_fuzz.py tests/test_gpu_edge_cases.py -m gpu -x -q 2>&1 | tail -3; fi
done
unset PIGEON_HIP_LIB
