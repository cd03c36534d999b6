"""Diagnostic: cold step and closed loop on every reference path (B = 4096, config-2 style inputs): phase times, share of instances served without the interior
point, stragglers."""
import os, sys, time, glob
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import _load_pkg
pkg = _load_pkg()
B = 4096
names = sorted(os.path.splitext(os.path.basename(f))[0] for f in glob.glob(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "paths", "*.npz")))
for path in names:
    traj = pkg.load_path_fixture(path)
    s_end = float(traj.s[-1])
    kw = dict(s_range=(2.0, 0.4 * s_end)) if s_end <= 90 else {}
    try:
        state, control, t0, toff = pkg.synthetic.config2_inputs(traj, B, seed=12345, **kw)
    except Exception as e:
        print(path, "inputs:", e); continue
    mpc = pkg.BatchedTrajectoryTrackingMPC(traj, B)
    mpc.set_inputs(state, control, t0, time_offset=toff)
    for _ in range(3):
        mpc.reset(); mpc.step_dev()
    mpc.synchronize(); a = time.perf_counter()
    for _ in range(10):
        mpc.reset(); mpc.step_dev()
    mpc.synchronize(); dt_cold = (time.perf_counter() - a) / 10
    st, it, _, _ = mpc.solve_info(); ps = mpc.polish_info(); ph = mpc.phase_ms()
    line = f"{path:14s} cold {1e3 * dt_cold:.3f} ms phases {[round(x, 3) for x in ph]} status {np.bincount(st, minlength=5).tolist()} iters>0 {(it > 0).sum()} (max {it.max()}) passes max {ps.max()}"
    mpc.simulate_(5); mpc.synchronize(); a = time.perf_counter(); mpc.simulate_(40); mpc.synchronize(); dt_cl = (time.perf_counter() - a) / 40
    st, it, _, _ = mpc.solve_info()
    print(line + f"; closed loop {1e3 * dt_cl:.3f} ms/step ({B / dt_cl / 1e6:.2f} M/s), last step status {np.bincount(st, minlength=5).tolist()} iters>0 {(it > 0).sum()} (max {it.max()})", flush=True)
    mpc.close()
