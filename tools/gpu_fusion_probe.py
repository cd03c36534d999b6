"""Diagnostic: cold and closed-loop step time with and without the fused step (pg_set_fusion: update_QP! inside the solve kernel)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import _load_pkg
pkg = _load_pkg()
import torch
for path in os.environ.get("PG_PATHS", "skidpadoval,vail,EastPaddock").split(","):
    traj = pkg.load_path_fixture(path)
    B = int(os.environ.get("PG_B", "4096"))
    state, control, t0, toff = pkg.synthetic.config2_inputs(traj, B, seed=12345)
    ref = None
    for ov in (0, 1, 2):
        mpc = pkg.BatchedTrajectoryTrackingMPC(traj, B)
        mpc.set_fusion(ov)
        mpc.set_inputs(state, control, t0, time_offset=toff)
        for _ in range(3):
            mpc.reset(); mpc.step_dev()
        mpc.synchronize(); a = time.perf_counter()
        for _ in range(20):
            mpc.reset(); mpc.step_dev()
        mpc.synchronize(); dt_cold = (time.perf_counter() - a) / 20
        u = mpc.get_next_control(); st, it, _, _ = mpc.solve_info()
        if ref is None: ref = u.copy()
        ph = mpc.phase_ms()
        mpc.simulate_(4); mpc.synchronize(); a = time.perf_counter(); mpc.simulate_(40); mpc.synchronize(); dt_cl = (time.perf_counter() - a) / 40
        print(f"{path} fusion mode {ov}: cold step {1e3 * dt_cold:.3f} ms ({B / dt_cold / 1e6:.2f} M solves/s) phases {[round(x, 3) for x in ph]}; closed loop {1e3 * dt_cl:.3f} ms/step ({B / dt_cl / 1e6:.2f} M/s); "
              f"solved {(st == 1).sum()}, identical controls: {np.array_equal(u, ref)}", flush=True)
        mpc.close()
