import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from conftest import load_pkg, make_oracle
from oracle import oracle as om, spec_numpy as S
pkg=load_pkg(); traj=pkg.load_path_fixture("vail"); orc=make_oracle(om,traj); P=S.X1(); U=S.coupled_control_params(); T=S.Trajectory(traj.data)
B=192
state, control, t0, toff = pkg.synthetic.config2_inputs(traj, B, seed=15, s_range=(5.0, float(traj.s[-1]) - 80.0))
state[:, 3] *= np.linspace(1.5, 4.5, B)
mpc = pkg.BatchedTrajectoryTrackingMPC(traj, B)
mpc.step_(state, control, t0, time_offset=toff)
qg, ug, pg_ = mpc.nodes()
for b in range(B):
    S.SATURATION_MARGINS=[]
    ts, dt = S.compute_time_steps(float(t0[b]))
    qs, us, ps = S.compute_linearization_nodes(P, U, T, state[b], control[b], ts, dt, 10, 20, time_offset=float(toff[b]))
    mg=np.array(S.SATURATION_MARGINS)
    d=np.abs(ug[b]-us)/np.maximum(1,np.abs(us)); dq=np.abs(qg[b]-qs)/np.maximum(1,np.abs(qs))
    if (d.max()>1e-7 or dq.max()>1e-7) and mg.min()>=1e-9:
        i=int(np.argmax(d[:,0])); print(b, "min margin %.2e"%mg.min(), "us diff %.2e at node %d"%(d.max(), i), "gpu", ug[b,i], "spec", us[i], "q diff %.2e"%dq.max(), "state", state[b], "ctrl", control[b])
b=147
ts, dt = S.compute_time_steps(float(t0[b]))
qs, us, ps = S.compute_linearization_nodes(P, U, T, state[b], control[b], ts, dt, 10, 20, time_offset=float(toff[b]))
np.set_printoptions(precision=10, linewidth=200)
print("node, gpu us, spec us, gpu ds, spec ds, gpu p, spec p")
for i in range(0, 14):
    print(i, ug[b,i], us[i], qg[b,i,0], qs[i,0], pg_[b,i,:2], ps[i,:2])
