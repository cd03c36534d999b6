"""Diagnostic (round 6): the lateral closed loop of the benchmark batch over STEPS steps on the device; the instances that end furthest from the path are run again through
the ORACLE loop (its nodes / update_QP / exact verified optimum / plant: tests/test_gpu_decoupled_closed_loop.py) from the same initial states -- is the drift the device
loop shows (|e| of 10-15 m on 5-10 % of the skidpad batch after 250 steps, tools/gpu_soak.py with PG_FORM=dec) the formulation's own behaviour?
usage: tools/gpu_lat_drift.py [--steps 300] [--walls] [--n 8]"""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import _load_pkg
pkg = _load_pkg()
from oracle import oracle as oracle_mod
from test_gpu_decoupled_closed_loop import oracle_lateral_loop
ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=300); ap.add_argument("--walls", action="store_true"); ap.add_argument("--n", type=int, default=8)
ap.add_argument("--path", default="skidpadoval")
a = ap.parse_args()
traj = pkg.load_path_fixture(a.path)
B, Ns, Nl, Ww = 4096, 10, 40, 1000.0
state, control, t0, toff = pkg.synthetic.config2_inputs(traj, B, seed=12345)
mpc = pkg.DecoupledTrajectoryTrackingMPC(pkg.X1(), traj, B, N_short=Ns, N_long=Nl, walls=a.walls, wall_weight=Ww)
mpc.set_inputs(state, control, t0, time_offset=toff)
s, c, t, qh, uh = mpc.simulate_(a.steps, dt=0.01, record=True)
st, it, act, mu = mpc.solve_info()
sep = mpc.path_coordinates()
e = np.abs(sep[:, 1])
print(f"device loop, {a.steps} steps: status {np.bincount(st, minlength=6).tolist()}, |e| > 1 m: {(e > 1).sum()}, > 5 m: {(e > 5).sum()}, max {np.nanmax(e):.2f} m")
worst = np.argsort(-np.nan_to_num(e))[:a.n]
calm = np.argsort(np.nan_to_num(e, nan=1e9))[:a.n // 2]
sel = np.concatenate([worst, calm])
print("instances (worst first, then calm):", sel.tolist(), " final |e| on the device:", np.round(e[sel], 2).tolist())
oq, ou, qf, uf, ok = oracle_lateral_loop(oracle_mod, traj, Ns, Nl, a.walls, Ww, state[sel], control[sel], t0[sel], toff[sel], a.steps)
print("oracle loop: every step a verified optimum:", ok.tolist())
print("distance between the final positions of the two loops [m]:", np.round(np.hypot(s[sel, 0] - qf[:, 0], s[sel, 1] - qf[:, 1]), 2).tolist())
for k in (10, 40, 100, 200, a.steps - 1):
    if k < a.steps:
        d = np.max(np.abs(qh[k][sel] - oq[k]) / np.maximum(1.0, np.abs(oq[k])), axis=1)
        print(f"  step {k}: max relative state difference device vs oracle per instance: {np.array2string(d, precision=1)}")
