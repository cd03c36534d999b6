"""Diagnostic: closed-loop steps on the device (pg_simulate_dev) with and without the warm start of the active set -- wall time per step, share of instances
whose warm polish verifies (iters == 0), polish rounds."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import _load_pkg
pkg = _load_pkg()
traj = pkg.load_path_fixture("skidpadoval")
B = 4096
state, control, t0, toff = pkg.synthetic.config2_inputs(traj, B, seed=12345)
for warm in (True, False):
    mpc = pkg.BatchedTrajectoryTrackingMPC(traj, B, warm_polish=warm)
    mpc.set_inputs(state, control, t0, time_offset=toff)
    mpc.simulate_(1)                                          # cold step + plant advance
    mpc.simulate_(3)
    a = time.perf_counter(); mpc.simulate_(40); dtw = time.perf_counter() - a
    st, it, _, _ = mpc.solve_info(); ps = mpc.polish_info()
    print(f"warm_polish={warm}: {1e3 * dtw / 40:.3f} ms per closed-loop step ({B * 40 / dtw / 1e6:.2f} M solves/s); last step: solved {(st == 1).sum()}, warm-verified (iters == 0) {(it == 0).sum()}, "
          f"iters mean {it.mean():.2f}, polish rounds {np.bincount(ps + 1, minlength=8).tolist()}", flush=True)
    mpc.close()
