cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r6
python tools/gpu_lat_handover.py --walls --settings "1,lat_hand_target=1500;1,lat_hand_target=1500,nodes_serial=1;0" 2>&1 | grep -v amdgpu.ids | cut -c1-330
python -m pytest tests/test_gpu_decoupled.py tests/test_gpu_decoupled_closed_loop.py tests/test_gpu_parity.py tests/test_gpu_full_size.py tests/test_gpu_api_contract.py -x -q > gpurun_out/r6/gputests_4.txt 2>&1; tail -12 gpurun_out/r6/gputests_4.txt
