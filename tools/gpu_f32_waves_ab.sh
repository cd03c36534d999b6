# fp32 A/B of an alternative build of the fp32 library (libpigeon_hip_f32_w1.so: whatever experiment it was built with) against the shipped one: config 3, 4096 without the row, 8192 per GPU
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B="bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-decoupled --no-rollout --no-warm"
run() {
  timeout -k 10 300 python $B > gpurun_out/bench_c3_$1.log 2>&1
  tail -1 gpurun_out/bench_c3_$1.log | python -c "
import sys,json; d=json.loads(sys.stdin.read()); f=d['fp32']; w=f['without_hji']
print('$1 config3', round(f['value']), [round(x,4) for x in f['phase_ms']], f['solved'], f.get('polished'), 'without_hji', (round(w['value']), [round(x,4) for x in w['phase_ms']]) if isinstance(w,dict) else w)"
  timeout -k 10 300 python $B --no-hji --no-f32 --precision f32 --batch 8192 > gpurun_out/bench_f32_8192_$1.log 2>&1
  tail -1 gpurun_out/bench_f32_8192_$1.log | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$1 f32 8192', round(d['value']), [round(x,4) for x in d['phase_ms'].values()], d['solved'], d['ipm_iters_hist'], d['polish_rounds_hist'])"
}
set -e
run base
test -f pigeon.jl_amd/csrc/libpigeon_hip_f32_w1.so      # (nothing to compare against otherwise)
export PIGEON_HIP_LIB_F32=$PWD/pigeon.jl_amd/csrc/libpigeon_hip_f32_w1.so      # selected, never copied over the shipped library (pigeon.jl_amd/_lib.py)
run ext
timeout -k 10 400 python -m pytest tests/test_gpu_f32.py -m gpu -x -q 2>&1 | tail -3
unset PIGEON_HIP_LIB_F32
