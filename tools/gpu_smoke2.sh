cd $GRAFT_REPO_ROOT
cat > /tmp/dbg.py <<'PY'
import sys, numpy as np, time, os
np.set_printoptions(precision=6, suppress=True, linewidth=200)
sys.path.insert(0,'tests')
from conftest import load_pkg, make_oracle
pkg=load_pkg(); sk=pkg.load_path_fixture('skidpadoval')
B=8
mpc=pkg.BatchedTrajectoryTrackingMPC(sk,B)
state,control,t0,toff=pkg.synthetic.config2_inputs(sk,B,seed=12345)
for rep in range(3):
    mpc.reset()
    mpc.set_inputs(state,control,t0,time_offset=toff); mpc.compute_time_steps_(); mpc.compute_linearization_nodes_()
    qs,us,ps=mpc.nodes()
    print(os.environ.get('HIP_LAUNCH_BLOCKING'), 'rep',rep,'gpu us[0]', us[:3,0].ravel(), 'expected', control[:3,0], (control[:3,1]+control[:3,2]))
PY
python /tmp/dbg.py 2>&1 | grep rep
HIP_LAUNCH_BLOCKING=1 python /tmp/dbg.py 2>&1 | grep rep
