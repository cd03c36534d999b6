cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r6
python -m pytest tests/test_gpu_f32.py "tests/test_gpu_decoupled.py" -x -q -s > gpurun_out/r6/gputests_5.txt 2>&1; grep -v "amdgpu.ids" gpurun_out/r6/gputests_5.txt | tail -25
