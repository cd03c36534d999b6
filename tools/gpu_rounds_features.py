"""Dump, for the cold 4096 headline batch, the active-set rounds every instance needed next to cheap features of its inputs (for the launch order of k_solve):
gpurun_out/rounds_features.npz.  python tools/gpu_rounds_features.py [path]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge._load_pkg()
name = sys.argv[1] if len(sys.argv) > 1 else "skidpadoval"
traj = pkg.load_path_fixture(name)
B = 4096
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 12345
kw = dict(s_range=(2.0, 0.4 * float(traj.s[-1]))) if float(traj.s[-1]) <= 100 else {}
state, control, t0, toff = pkg.synthetic.config2_inputs(traj, B, seed=seed, **kw)
mpc = pkg.BatchedTrajectoryTrackingMPC(traj, B)
u, status, iters = mpc.step_(state, control, t0, time_offset=toff)
qs, us, ps = mpc.nodes(); sep = mpc.path_coordinates(); pol = mpc.polish_info(); st, it, act, mu = mpc.solve_info()
nact = np.array([[bin(int(m)).count("1") for m in row] for row in act])
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", f"rounds_features_{name}_{seed}.npz"), state=state, control=control, qs=qs, us=us, ps=ps, sep=sep, pol=pol, iters=it, nact=nact, u=u, act=act)
print("rounds hist", np.bincount(np.maximum(pol, 0)), "active rows mean", nact.sum(1).mean())
