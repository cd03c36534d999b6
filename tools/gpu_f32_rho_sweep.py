"""fp32 cold batch (8192 instances, config 4's per-GPU share): solve time, instances that reach the interior point and accuracy-neutral counters against
pg_config.polish_rho / polish_tol.   python tools/gpu_f32_rho_sweep.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge._load_pkg()
traj = pkg.load_path_fixture("skidpadoval")
B = 8192
state, control, t0, toff = pkg.synthetic.config2_inputs(traj, B, seed=12345)
m64 = pkg.BatchedTrajectoryTrackingMPC(traj, B)                        # the fp64 library on the same inputs: the reference for the accuracy columns
m64.step_(state, control, t0, time_offset=toff); u64 = m64.get_next_control().copy(); m64.close()
ref = None
for rho, tol in ((1e3, 1e-4), (1e4, 1e-4), (3e4, 1e-4), (1e5, 1e-4), (3e5, 1e-4), (1e6, 1e-4), (3e4, 3e-5), (1e5, 3e-5), (1e5, 1e-5)):
    mpc = pkg.BatchedTrajectoryTrackingMPC(traj, B, precision="f32", polish_rho=rho, polish_tol=tol)
    mpc.set_inputs(state, control, t0, time_offset=toff)
    for _ in range(3):
        mpc.reset(); mpc.step_dev()
    mpc.synchronize(); ph = []
    for _ in range(10):
        mpc.reset(); mpc.step_dev(); mpc.synchronize(); ph.append(mpc.phase_ms())
    ph = np.mean(np.array(ph), axis=0)
    st, it, act, mu = mpc.solve_info(); pol = mpc.polish_info(); u = mpc.get_next_control().copy()
    if ref is None:
        ref = u
    un = np.asarray(mpc.u_normalization, dtype=float)
    du = np.abs(u - ref) / np.array([un[0], un[1], un[1]])
    d64 = (np.abs(u - u64) / np.array([un[0], un[1], un[1]]))[:, 0]
    print(f"rho {rho:7.0f} tol {tol:.0e}: solve {ph[2]:.3f} ms  solved {int((st == pkg.SOLVED).sum())}/{B}  interior point for {int((it > 0).sum())}  unverified {int((pol < 0).sum())}  rounds mean {pol[pol > 0].mean():.2f} max {pol.max()}"
          f"  |u - u(rho=1e3)| max {du.max():.1e}  vs fp64 library (delta): median {np.median(d64):.1e} p99 {np.percentile(d64, 99):.1e} max {d64.max():.1e}", flush=True)
    mpc.close()
