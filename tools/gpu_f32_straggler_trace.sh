# fp32 cold batch on skidpadoval: which instances leave the rounds from the empty set for the interior point, and the polish trace of the two slowest
cd $GRAFT_REPO_ROOT
export PG_PREC=f32 PG_PATH=skidpadoval PG_STEP=1
PG_DEBUG_INSTANCE=0 timeout -k 10 120 python tools/gpu_polish_trace.py 2>&1 | grep "slowest" > gpurun_out/f32_slowest.txt
cat gpurun_out/f32_slowest.txt
for i in $(python -c "
import re; s=open('gpurun_out/f32_slowest.txt').read(); print(' '.join(m for m in re.findall(r'\((\d+), \d+, -?\d+\)', s)[:2]))"); do
  echo "=== instance $i"; PG_DEBUG_INSTANCE=$i timeout -k 10 120 python tools/gpu_polish_trace.py 2>&1 | grep -v amdgpu.ids | tail -45
done
