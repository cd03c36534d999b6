# one call: smoke, full default bench line, GPU test suite, then the rocprofv3 campaign of tools/gpu_profile.sh
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 &&
timeout -k 10 400 python bench.py > gpurun_out/bench_full.log 2> gpurun_out/bench_full.err &&
tail -c 600 gpurun_out/bench_full.log &&
(timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/pytest.log 2>&1; tail -3 gpurun_out/pytest.log) &&
bash tools/gpu_profile.sh
