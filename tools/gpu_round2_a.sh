cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
PG_CASES="1e-12:0,1e-8:1,1e-6:1,1e-5:1,1e-4:1,1e-3:1,1e-2:1" python tools/gpu_accuracy_full.py > gpurun_out/acc2.log 2>&1; cat gpurun_out/acc2.log
PG_PREC=f32 PG_CASES="1e-5:0,1e-5:1,1e-4:1,1e-3:1,1e-2:1" python tools/gpu_accuracy_full.py > gpurun_out/acc2_f32.log 2>&1; cat gpurun_out/acc2_f32.log
PG_PREC=f32 PG_RHO=1e4 PG_CASES="1e-4:1,1e-3:1" python tools/gpu_accuracy_full.py > gpurun_out/acc2_f32b.log 2>&1; cat gpurun_out/acc2_f32b.log
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-hji --no-decoupled > gpurun_out/bench2.log 2>&1
tail -1 gpurun_out/bench2.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ['value','ms_per_step','phase_ms','warm_value','solved','ipm_iters_mean']}); print(d.get('fp32'))"
