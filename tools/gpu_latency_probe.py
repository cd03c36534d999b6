"""Diagnostic: end-to-end step latency (host call to controls on the host) for small batches, both precisions."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import _load_pkg
pkg = _load_pkg()
traj = pkg.load_path_fixture("skidpadoval")
for prec in ["f64", "f32"]:
    for B in [1, 8, 64, 512]:
        state, control, t0, toff = pkg.synthetic.config2_inputs(traj, B, seed=3)
        m = pkg.BatchedTrajectoryTrackingMPC(traj, B, precision=prec)
        for _ in range(5):
            m.step_(state, control, t0, time_offset=toff)
        ts = []
        for _ in range(50):
            a = time.perf_counter(); m.step_(state, control, t0 + 0.01, time_offset=toff); ts.append(time.perf_counter() - a)
        print(f"{prec} B={B:4d}: pg_step (host in, host out) median {1e3*np.median(ts):.3f} ms  min {1e3*min(ts):.3f} ms  device phases {np.round(m.phase_ms(), 3)}", flush=True)
        m.close()
