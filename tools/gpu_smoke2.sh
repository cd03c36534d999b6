cd $GRAFT_REPO_ROOT
python - <<'PY' > gpurun_out/dbg.log 2>&1
import sys, numpy as np
sys.path.insert(0,'tests')
from conftest import load_pkg
pkg=load_pkg(); sk=pkg.load_path_fixture('skidpadoval')
for (Ns,Nl) in [(10,20),(10,40)]:
    B=4096
    mpc=pkg.DecoupledTrajectoryTrackingMPC(pkg.X1(), sk, B, N_short=Ns, N_long=Nl)
    state,control,t0,toff=pkg.synthetic.config2_inputs(sk,B,seed=31,traj_mode=False)
    u,st,it=mpc.step_(state,control,t0,time_offset=toff)
    s2,i2,act,mu=mpc.solve_info()
    print(Ns,Nl,'iters mean',it.mean(),'max',it.max(),'status counts',np.bincount(st), 'mu max', mu.max(), 'n(mu>1e-11)', int((mu>1e-11).sum()))
mpc=pkg.BatchedTrajectoryTrackingMPC(sk, 4096)
state,control,t0,toff=pkg.synthetic.config2_inputs(sk,4096,seed=12345)
u,st,it=mpc.step_(state,control,t0,time_offset=toff); s2,i2,act,mu=mpc.solve_info(); print('coupled iters', it.mean(), it.max(), np.bincount(st), 'mu max', mu.max())
PY
tail -4 gpurun_out/dbg.log
