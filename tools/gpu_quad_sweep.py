"""Diagnostic: solve-phase time of both QP kernels as a function of the batch size."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import _load_pkg
pkg = _load_pkg()
traj = pkg.load_path_fixture("skidpadoval")
for solver in ["wave", "quad"]:
    os.environ["PG_SOLVER"] = solver
    for B in [256, 1024, 2048, 4096, 8192, 16384]:
        mpc = pkg.BatchedTrajectoryTrackingMPC(traj, B)
        state, control, t0, toff = pkg.synthetic.config2_inputs(traj, B, seed=12345)
        ms = []
        for rep in range(4):
            mpc.reset()
            mpc.step_(state, control, t0, time_offset=toff)
            ms.append(mpc.phase_ms()[2])
        st, it, act, mu = mpc.solve_info()
        print(f"{solver} B={B:6d} solve {min(ms):8.3f} ms  {B/min(ms)/1e3:8.1f} k solves/ms... iters mean {it.mean():.2f} max {it.max()} solved {(st==1).sum()}", flush=True)
        mpc.close()
