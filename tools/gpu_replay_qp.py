"""Diagnostic: solve recorded QP data (gpurun_out/fuzz_dump_*.npz of tools/gpu_fuzz_full.py, or any [qp_len] block) with k_solve through pg_set_qp + pg_solve and compare
with the oracle's exact solver: distance of the controls, objective gap, feasibility.  PG_REPLAY="file,file,..."; PG_RHO, PG_CG (cold_guess), PG_PTOL, PG_POLISH override the config."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import _load_pkg
pkg = _load_pkg()
from oracle import oracle as oracle_mod
files = os.environ["PG_REPLAY"].split(",")
traj = pkg.load_path_fixture("skidpadoval")
kw = {}
if os.environ.get("PG_RHO"): kw["polish_rho"] = float(os.environ["PG_RHO"])
if os.environ.get("PG_CG"): kw["cold_guess"] = int(os.environ["PG_CG"])
if os.environ.get("PG_PTOL"): kw["polish_tol"] = float(os.environ["PG_PTOL"])
if os.environ.get("PG_POLISH"): kw["polish"] = bool(int(os.environ["PG_POLISH"]))
B = len(files)
mpc = pkg.BatchedTrajectoryTrackingMPC(traj, max(B, 2), **kw)
state, control, t0, toff = pkg.synthetic.config2_inputs(traj, max(B, 2), seed=1)
mpc.step_(state, control, t0, time_offset=toff)                 # (any valid step: sizes the batch)
mpc.reset()
qps = np.stack([np.load(f)["qp"] for f in files] + ([np.load(files[0])["qp"]] if B < 2 else []))
mpc.set_qp_data(qps)
mpc.solve_()
x, _ = mpc.solution(); st, it, act, mu = mpc.solve_info(); pol = mpc.polish_info()
orc = oracle_mod.Oracle()
for i, f in enumerate(files):
    xe, ye, info = orc.solve_exact(qps[i]); S = orc.split_x(xe)
    err1 = np.max(np.abs(x[i, 1, 6:] - S["u"][1])); erra = np.max(np.abs(x[i, :, 6:] - S["u"]))
    print(f"{os.path.basename(f)}: status {st[i]} iters {it[i]} polish {pol[i]}; oracle status {info['status']} iters {info['iters']}; applied control off by {err1:.3g}, any control {erra:.3g}, states {np.max(np.abs(x[i, :, :6] - S['q'])):.3g}")
mpc.close()
