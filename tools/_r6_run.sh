cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r6
python -m pytest tests/test_gpu_f32.py tests/test_gpu_abi_smoke.py "tests/test_gpu_full_size.py::test_closed_loop_on_device_matches_oracle" "tests/test_gpu_decoupled_closed_loop.py::test_decoupled_closed_loop_on_device_matches_oracle" tests/test_gpu_onvehicle.py -q > gpurun_out/r6/gputests_3.txt 2>&1
tail -4 gpurun_out/r6/gputests_3.txt
hipcc --offload-arch=gfx950 -O3 tools/probes/accvgpr_count_probe.hip -o /tmp/acc_probe 2>/dev/null
rocprofv3 --pmc SQ_INSTS_VALU --output-format csv -d gpurun_out/r6/accprobe -- /tmp/acc_probe > gpurun_out/r6/accprobe.log 2>&1
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("gpurun_out/r6/accprobe/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
for k,v in acc.items(): print(k, v)
PY
python bench.py --no-cpu-baseline > gpurun_out/r6/bench_a.log 2>&1; tail -c 3000 gpurun_out/r6/bench_a.log
