import os, sys
import numpy as np
sys.path.insert(0, "/root/repo")
from __graft_entry__ import _load_pkg
pkg = _load_pkg()
traj = pkg.load_path_fixture("skidpadoval")
n = 4096
state, control, t0, toff = pkg.synthetic.config2_inputs(traj, n, seed=8)
state = state.astype(np.float32).astype(np.float64); control = control.astype(np.float32).astype(np.float64)
d64 = pkg.DecoupledTrajectoryTrackingMPC(pkg.X1(), traj, n, N_short=10, N_long=40)
u64, st64, it64 = d64.step_(state, control, t0, time_offset=toff)
print("f64 solved", (st64 == 1).sum(), "iters", it64.mean(), it64.max())
for tol in [1e-3, 1e-4, 1e-5]:
    for mi in [40]:
        d32 = pkg.DecoupledTrajectoryTrackingMPC(pkg.X1(), traj, n, N_short=10, N_long=40, precision="f32", ipm_tol=tol, ipm_max_iter=mi)
        u32, st32, it32 = d32.step_(state, control, t0, time_offset=toff)
        ok = (st32 == 1) & (st64 == 1)
        err = np.abs(u32[ok, 0] - u64[ok, 0]) / 0.314159
        print(f"tol {tol:g} maxit {mi}: status {np.bincount(st32, minlength=5).tolist()} iters mean {it32.mean():.1f} max {it32.max()} err median {np.median(err):.2e} p99 {np.percentile(err, 99):.2e} max {err.max():.2e}  ms {d32.phase_ms()}")
        d32.close()
