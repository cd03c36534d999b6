"""Diagnostic (round 6): do the two arrangements of k_solve_lat (four instances per wavefront with the row state in the workspace / one instance per wavefront with it in
registers) produce the same BITS?  Interior point only (polish off), iteration cap 1, 2, 3, ...: the damped iterate pg_get_solution returns after k iterations, the
multipliers, mu -- first k at which they differ and on how many instances.   usage: tools/gpu_lat_arrangements.py [--walls] [--batch 256]"""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import _load_pkg
pkg = _load_pkg()
ap = argparse.ArgumentParser(); ap.add_argument("--walls", action="store_true"); ap.add_argument("--batch", type=int, default=256); ap.add_argument("--caps", default="1,2,3,5,8")
a = ap.parse_args()
traj = pkg.load_path_fixture("skidpadoval")
B = a.batch
state, control, t0, toff = pkg.synthetic.config2_inputs(traj, B, seed=12345)
for cap in [int(x) for x in a.caps.split(",")]:
    out = []
    for opts in ({"lat_single_max": 1 << 20, "lat_handover": 0}, {"lat_single_max": 0, "lat_handover": 0}):
        m = pkg.DecoupledTrajectoryTrackingMPC(pkg.X1(), traj, B, N_short=10, N_long=40, walls=a.walls, polish=False, ipm_max_iter=cap, options=opts)
        u, st, it = m.step_(state, control, t0, time_offset=toff)
        x, sg = m.solution(); lam = m.multipliers(); _, _, _, mu = m.solve_info()
        out.append((x.copy(), sg.copy(), lam.copy(), mu.copy(), it.copy(), m.get_option("stat_lat_one_per_wavefront_solves")))
        m.close()
    (x1, s1, l1, m1, i1, n1), (x4, s4, l4, m4, i4, n4) = out
    dx = (x1 != x4).reshape(B, -1).any(axis=1); ds = (s1 != s4).reshape(B, -1).any(axis=1); dl = (l1 != l4).reshape(B, -1).any(axis=1); dm = m1 != m4
    print(f"cap {cap} (one-per-wavefront launches {int(n1)}/{int(n4)}): x differs on {int(dx.sum())}, sigma on {int(ds.sum())}, multipliers on {int(dl.sum())}, mu on {int(dm.sum())} of {B}; "
          f"max |dx| {np.nanmax(np.abs(x1 - x4)):.2e}, max rel |dmu| {np.nanmax(np.abs(m1 - m4) / np.maximum(np.abs(m4), 1e-300)):.2e}", flush=True)
    if dx.any() or dm.any():
        b = int(np.argmax(dx | dm)); w = np.argwhere(x1[b] != x4[b])
        print(f"   first instance {b}: mu {m1[b]!r} / {m4[b]!r}; x differs at (node, component) {w[:6].tolist()}")
