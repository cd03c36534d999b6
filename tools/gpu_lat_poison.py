"""Diagnostic (round 6, -DPG_DIAG library): k_solve_lat with its workspaces and its LDS filled with NaN before every launch (option "diag_lat_poison") against the same
calls without: any difference is a read of memory the launch has not written itself -- stale data of whichever instance last used that block (timing-dependent in the
list-mode launches, whose blocks serve the instances in the order the to-do list was filled).   usage: tools/gpu_lat_poison.py [--walls] [--steps 6]"""
import argparse, os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import _load_pkg
pkg = _load_pkg()
ap = argparse.ArgumentParser(); ap.add_argument("--walls", action="store_true"); ap.add_argument("--steps", type=int, default=6); ap.add_argument("--batch", type=int, default=4096)
a = ap.parse_args()
traj = pkg.load_path_fixture("skidpadoval")
B = a.batch
state, control, t0, toff = pkg.synthetic.config2_inputs(traj, B, seed=12345)
for opts in ({"lat_handover": 0, "lat_single_max": 0, "lat_split": 0}, {"lat_handover": 0, "lat_single_max": 0}, {"lat_single_max": 0}, {}, {"lat_split": 0}):
    ms = [pkg.DecoupledTrajectoryTrackingMPC(pkg.X1(), traj, B, N_short=10, N_long=40, walls=a.walls, precision="f64-diag", options=dict(opts, **extra)) for extra in ({}, {"diag_lat_poison": 1})]
    for m in ms: m.set_inputs(state, control, t0, time_offset=toff)
    for k in range(a.steps):
        out = []
        for m in ms:
            s, c, t = m.simulate_(1)[:3]
            st, it, act, mu = m.solve_info(); x, sg = m.solution()
            out.append((np.asarray(c).copy(), st.copy(), it.copy(), x.copy(), m.polish_info().copy()))
        dx = np.where((out[0][3] != out[1][3]).reshape(B, -1).any(axis=-1) | (out[0][1] != out[1][1]) | (out[0][2] != out[1][2]))[0]
        nanx = int(np.isnan(out[1][3]).reshape(B, -1).any(axis=-1).sum())
        print(f"{opts} step {k}: poisoned run differs on {len(dx)} instances (status hist clean {np.bincount(out[0][1], minlength=6).tolist()} poisoned {np.bincount(out[1][1], minlength=6).tolist()}, NaN solutions {nanx})", flush=True)
        if len(dx):
            for b in dx[:6]: print(f"    instance {b}: status {out[0][1][b]}/{out[1][1][b]} iters {out[0][2][b]}/{out[1][2][b]} polish {out[0][4][b]}/{out[1][4][b]} max|dx| {np.nanmax(np.abs(out[0][3][b] - out[1][3][b])):.2e}")
            break
    for m in ms: m.close()
