"""Diagnostic (round 6): k_solve_lat at the ends of its horizon range (N = 63, 17, 15: workspace / register variants, hand-over, one per wavefront) against the four-per-wavefront single launch and the embedding in k_solve."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import _load_pkg
pkg = _load_pkg()
traj = pkg.load_path_fixture("skidpadoval")
for (Ns, Nl, B, walls) in ((10, 53, 300, True), (10, 53, 1300, False), (5, 12, 700, True), (10, 7, 1100, False), (3, 14, 64, True)):
    state, control, t0, toff = pkg.synthetic.config2_inputs(traj, B, seed=5)
    out = {}
    for name, opts in (("default", {"lateral_solver": 1}), ("four", {"lateral_solver": 1, "lat_handover": 0, "lat_single_max": 0}), ("embed", {"lateral_solver": 2})):
        try:
            m = pkg.DecoupledTrajectoryTrackingMPC(pkg.X1(), traj, B, N_short=Ns, N_long=Nl, walls=walls, options=opts)
        except Exception as e:
            print(name, "refused:", str(e)[:100]); continue
        u, st, it = m.step_(state, control, t0, time_offset=toff)
        out[name] = (u.copy(), st.copy(), m.polish_info().copy(), m.get_option("stat_lat_handover_solves"), m.get_option("stat_lat_one_per_wavefront_solves"), m.get_option("lateral_solver_in_use"))
        m.close()
    ud, sd, pd, hd, od, _ = out["default"]
    line = f"N={Ns+Nl} B={B} walls={walls}: default solved {int(pkg.is_solved(sd).sum())}/{B} verified {int((pd>=1).sum())} (hand-over {int(hd)}, one-per-wavefront {int(od)})"
    for other in ("four", "embed"):
        if other in out:
            uo, so, po = out[other][:3]; both = (pd >= 1) & (po >= 1)
            line += f"; vs {other}: solved {int(pkg.is_solved(so).sum())}, both verified {int(both.sum())}, max |du| {np.max(np.abs(ud[both,0]-uo[both,0])) if both.any() else float('nan'):.1e}"
    print(line, flush=True)
