cd $GRAFT_REPO_ROOT
export PG_PREC=f64 PG_PATH=skidpadoval PG_STEP=1
python - <<'P' > gpurun_out/f64_rounds.txt 2>&1
import os, sys, numpy as np
sys.path.insert(0, '.')
from __graft_entry__ import _load_pkg
pkg = _load_pkg()
traj = pkg.load_path_fixture("skidpadoval")
B=4096
state, control, t0, toff = pkg.synthetic.config2_inputs(traj, B, seed=12345)
mpc = pkg.BatchedTrajectoryTrackingMPC(traj, B)
mpc.step_(state, control, t0, time_offset=toff)
ps = mpc.polish_info()
for r in (2,3,4,5):
    print(r, np.where(ps==r)[0][:6].tolist())
P
cat gpurun_out/f64_rounds.txt
for i in $(python -c "
import re; L=open('gpurun_out/f64_rounds.txt').read().split('\n'); 
import ast
out=[]
for l in L:
    if l[:1] in '2345' and '[' in l: out += ast.literal_eval(l[2:])[:3]
print(' '.join(map(str,out)))"); do
  echo "=== instance $i"; PG_DEBUG_INSTANCE=$i timeout -k 10 120 python tools/gpu_polish_trace.py 2>&1 | grep -A12 "polish checks" | tail -12
done
