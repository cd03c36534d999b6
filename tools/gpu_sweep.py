"""Sweep of interior-point parameters on the GPU: iterations and accuracy against the oracle's exact optimum (diagnostic)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import load_pkg, make_oracle
from oracle import oracle as om
pkg = load_pkg(); sk = pkg.load_path_fixture("skidpadoval")
B = 192
state, control, t0, toff = pkg.synthetic.config2_inputs(sk, B, seed=12345)
orc = make_oracle(om, sk)
ref = None
for mu0 in [10.0, 100.0, 1000.0]:
    for tol in [1e-13, 1e-12, 1e-11, 1e-10, 1e-9]:
        mpc = pkg.BatchedTrajectoryTrackingMPC(sk, B, ipm_tol=tol, ipm_mu0=mu0)
        u, st, it = mpc.step_(state, control, t0, time_offset=toff)
        x, _ = mpc.solution()
        if ref is None:
            qp = mpc.qp_data()
            ref = np.array([orc.split_x(orc.solve_exact(qp[b])[0])["u"] for b in range(B)])
        e2 = np.abs(x[:, 1, 6:] - ref[:, 1]).max(); eall = np.abs(x[:, :, 6:] - ref).max()
        print(f"mu0 {mu0:7.1f} tol {tol:.0e} iters mean {it.mean():5.2f} max {it.max():2d} solved {int((st==1).sum())}/{B} err(u2) {e2:.2e} err(all u) {eall:.2e}", flush=True)
