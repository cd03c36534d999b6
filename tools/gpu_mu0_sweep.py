"""Diagnostic: interior-point iterations / solve time of the cold config-2 batch against the initial barrier parameter mu0 and the hand-over tolerance."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import _load_pkg
pkg = _load_pkg(); traj = pkg.load_path_fixture("skidpadoval"); B = 4096
state, control, t0, toff = pkg.synthetic.config2_inputs(traj, B, seed=12345)
for mu0 in [float(x) for x in os.environ.get("PG_MU0", "1,3,10,30,100,300,1000").split(",")]:
    for tol in [float(x) for x in os.environ.get("PG_TOLS", "1e-6").split(",")]:
        mpc = pkg.BatchedTrajectoryTrackingMPC(traj, B, ipm_mu0=mu0, polish_ipm_tol=tol)
        ms = []
        for _ in range(4):
            mpc.reset(); u, st, it = mpc.step_(state, control, t0, time_offset=toff); ms.append(mpc.phase_ms()[2])
        ps = mpc.polish_info()
        print(f"mu0 {mu0:g} tol {tol:g}: solve {min(ms):.3f} ms, iters mean {it.mean():.2f} max {it.max()}, solved {(st == 1).sum()}, polish rounds mean {np.mean(np.where(ps > 0, ps, 6)):.2f} failed {(ps < 0).sum()}", flush=True)
        mpc.close()
