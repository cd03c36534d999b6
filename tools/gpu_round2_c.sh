cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/bench_full.log 2> gpurun_out/bench_full.err; tail -c 6000 gpurun_out/bench_full.log; tail -5 gpurun_out/bench_full.err
bash tools/gpu_profile.sh
