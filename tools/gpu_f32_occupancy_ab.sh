# What does a second wave per SIMD buy k_solve?  The fp32 kernel is built for two (PG_F32_WAVES=2, shipped) and for one (libpigeon_hip_f32_w1.so: -DPG_F32_WAVES=1);
# same source, same batch, no stragglers since the settle rule of round 3.  Solve-phase time at 4096 / 8192 / 16384 cold instances.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
B="bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-decoupled --no-rollout --no-warm --no-hji --no-f32 --precision f32"
for lib in shipped w1; do
  if [ $lib = w1 ]; then export PIGEON_HIP_LIB_F32=$PWD/pigeon.jl_amd/csrc/libpigeon_hip_f32_w1.so; fi
  for n in 4096 8192 16384; do
    timeout -k 10 300 python $B --batch $n 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$lib', $n, 'solves/s', round(d['value']), 'phases', [round(x,4) for x in d['phase_ms'].values()], d['solved'], 'ipm', d['ipm_iters_hist'])"
  done
done
