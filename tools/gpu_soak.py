"""Diagnostic: long closed loop (PG_STEPS steps of 10 ms, default 3000 = 30 s of driving) of a 4096 batch on every reference path: status counts, instances that needed
the interior point, step time and tracking error per block of 250 steps.  Looks for slow degradation (stragglers accumulating, instances that stop solving).
PG_FORM=dec | dec_walls: the lateral formulation instead (round 6)."""
import os, sys, time, glob
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import _load_pkg
pkg = _load_pkg()
B = int(os.environ.get("PG_B", "4096")); STEPS = int(os.environ.get("PG_STEPS", "3000")); BLOCK = 250
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
names = os.environ.get("PG_PATHS", "").split(",") if os.environ.get("PG_PATHS") else sorted(os.path.splitext(os.path.basename(f))[0] for f in glob.glob(os.path.join(root, "tests", "golden", "paths", "*.npz")))
for path in names:
    traj = pkg.load_path_fixture(path)
    s_end = float(traj.s[-1])
    kw = dict(s_range=(2.0, 0.4 * s_end)) if s_end <= 90 else {}
    state, control, t0, toff = pkg.synthetic.config2_inputs(traj, B, seed=12345, **kw)
    # PG_FORM=dec / dec_walls: the lateral formulation (N = 50: k_solve_lat -- cold hand-over at the first step, warm attempts + list-mode launches afterwards)
    form = os.environ.get("PG_FORM", "coupled")
    mpc = pkg.BatchedTrajectoryTrackingMPC(traj, B) if form == "coupled" else pkg.DecoupledTrajectoryTrackingMPC(pkg.X1(), traj, B, N_short=10, N_long=40, walls=form == "dec_walls")
    mpc.set_inputs(state, control, t0, time_offset=toff)
    print(f"{path}: path length {s_end:.0f} m", flush=True)
    near_end = np.zeros(B, dtype=bool)                       # instances whose horizon has reached the end of the path at some point (the reference extrapolates there)
    for blk in range(STEPS // BLOCK):
        mpc.synchronize(); a = time.perf_counter()
        q, u, t = mpc.simulate_(BLOCK)[:3]
        mpc.synchronize(); dt = (time.perf_counter() - a) / BLOCK
        st, it, _, _ = mpc.solve_info(); pol = mpc.polish_info()
        sep = mpc.path_coordinates() if hasattr(mpc, "path_coordinates") else None
        e = np.abs(sep[:, 1]) if sep is not None else np.zeros(1)
        past = (sep[:, 0] > s_end - 1.0).sum() if sep is not None else -1
        near_end |= ~(sep[:, 0] < s_end - 60.0)               # (NaN counts as near the end: it stays flagged)
        trouble = (st != 1) | (pol < 1) | ~np.isfinite(q).all(axis=1)
        print(f"  steps {BLOCK * blk:5d}-{BLOCK * (blk + 1):5d}: {1e3 * dt:.3f} ms/step, status {np.bincount(st, minlength=5).tolist()}, interior point {(it > 0).sum()} (max {it.max()}), unverified {(pol < 1).sum()}, "
              f"|e| max {np.nanmax(e):.2f} m, past the path end {past}, finite {np.isfinite(q).all()}; in trouble with the path end > 60 m ahead: {(trouble & ~near_end).sum()}", flush=True)
    mpc.close()
