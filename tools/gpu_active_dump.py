"""Diagnostic: the verified working sets of the headline batch (cold step), with what a guess could be made from -- nodes, current control, QP bounds.
Writes gpurun_out/active_dump.npz.  Usage (GPU box): python tools/gpu_active_dump.py [B]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import _load_pkg
pkg = _load_pkg()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
traj = pkg.load_path_fixture("skidpadoval")
mpc = pkg.BatchedTrajectoryTrackingMPC(traj, B)
state, control, t0, toff = pkg.synthetic.config2_inputs(traj, B, seed=12345, traj_mode=True)
u, st, it = mpc.step_(state, control, t0, time_offset=toff)
st, it, act, mu = mpc.solve_info()
pol = mpc.polish_info()
qs, us, ps = mpc.nodes()
x, sg = mpc.solution()
lam = mpc.multipliers()
qp = mpc.qp_data()
ts, dt, _ = mpc.time_steps()
os.makedirs("gpurun_out", exist_ok=True)
np.savez_compressed("gpurun_out/active_dump.npz", state=state, control=control, status=st, iters=it, act=act, pol=pol, qs=qs, us=us, ps=ps, x=x, lam=lam.astype(np.float32), dt=dt,
                    qp=qp.astype(np.float32), qp_len=mpc.qp_len, N=mpc.N, Ns=mpc.N_short)
print("rounds histogram", sorted(zip(*np.unique(pol, return_counts=True))), "solved", int(pkg.is_solved(st).sum()))
