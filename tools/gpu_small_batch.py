"""Diagnostic: step latency of small batches (B = 1 .. 2048): device phases of a cold and of a warm pg_step_dev, and pg_step host-to-host in closed loop."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import _load_pkg
pkg = _load_pkg()
traj = pkg.load_path_fixture("skidpadoval")
for B in [int(x) for x in os.environ.get("PG_BS", "1,16,64,256,512,1024,2048").split(",")]:
    state, control, t0, toff = pkg.synthetic.config2_inputs(traj, max(B, 2), seed=12345)
    state, control, t0, toff = state[:B], control[:B], t0[:B], toff[:B]
    mpc = pkg.BatchedTrajectoryTrackingMPC(traj, B)
    mpc.set_inputs(state, control, t0, time_offset=toff)
    pc, pw = [], []
    for _ in range(6):
        mpc.reset(); mpc.step_dev(); mpc.synchronize(); pc.append(mpc.phase_ms())
        mpc.step_dev(); mpc.synchronize(); pw.append(mpc.phase_ms())
    pc = np.median(np.array(pc[1:]), 0); pw = np.median(np.array(pw[1:]), 0)
    a = time.perf_counter()
    for k in range(50):
        u, st, it = mpc.step_(state, control, t0 + 0.01 * k, time_offset=toff)
    host = (time.perf_counter() - a) / 50
    print(f"B={B:5d}: cold phases {[round(float(x), 3) for x in pc]} = {pc.sum():.3f} ms; warm phases {[round(float(x), 3) for x in pw]} = {pw.sum():.3f} ms; pg_step host-to-host (warm) {1e3 * host:.3f} ms", flush=True)
    mpc.close()
