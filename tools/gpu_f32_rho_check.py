"""fp32 library, candidate values of pg_config.polish_rho: cold step and 40 closed-loop steps on three reference paths (4096 instances), and config 3 (HJI row on the
13x13x9^5 grid): solve time, instances that reach the interior point or end unverified, accuracy of the applied steering against the fp64 library on the same inputs.
    python tools/gpu_f32_rho_check.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge._load_pkg()
B = 4096
RHOS = [float(x) for x in os.environ.get("PG_RHOS", "1e3,2e4,3e4").split(",")]


def report(tag, mpc, u64=None):
    st, it, act, mu = mpc.solve_info(); pol = mpc.polish_info(); ph = mpc.phase_ms(); u = mpc.get_next_control()
    s = f"{tag}: solve {ph[2]:.3f} ms  solved {int((st == pkg.SOLVED).sum())}/{B}  interior point {int((it > 0).sum())}  unverified {int((pol < 0).sum())}  rounds max {pol.max()}"
    if u64 is not None:
        d = np.abs(u[:, 0] - u64[:, 0]) / mpc.u_normalization[0]
        s += f"  vs fp64: median {np.median(d):.1e} p99 {np.percentile(d, 99):.1e} max {d.max():.1e}"
    print(s, flush=True)


for name in ("skidpadoval", "vail", "EastPaddock"):
    traj = pkg.load_path_fixture(name)
    state, control, t0, toff = pkg.synthetic.config2_inputs(traj, B, seed=12345)
    m64 = pkg.BatchedTrajectoryTrackingMPC(traj, B); m64.step_(state, control, t0, time_offset=toff); u64 = m64.get_next_control().copy(); m64.close()
    for rho in RHOS:
        mpc = pkg.BatchedTrajectoryTrackingMPC(traj, B, precision="f32", polish_rho=rho)
        mpc.set_inputs(state, control, t0, time_offset=toff)
        for _ in range(3):
            mpc.reset(); mpc.step_dev()
        mpc.synchronize()
        report(f"{name:12s} rho {rho:6.0f} cold      ", mpc, u64)
        worst_ipm = 0; worst_unv = 0; bad = 0; tt = 0.0
        for k in range(40):
            t = time.perf_counter(); mpc.simulate_(1); mpc.synchronize(); tt += time.perf_counter() - t
            st, it, _, _ = mpc.solve_info(); pol = mpc.polish_info()
            worst_ipm = max(worst_ipm, int((it > 0).sum())); worst_unv = max(worst_unv, int((pol < 0).sum())); bad = max(bad, int((st != pkg.SOLVED).sum()))
        print(f"{name:12s} rho {rho:6.0f} closed loop: {tt / 40 * 1e3:.3f} ms per step (host-timed, one step per call)  worst step: interior point {worst_ipm}  unverified {worst_unv}  not solved {bad}", flush=True)
        mpc.close()

traj = pkg.load_path_fixture("skidpadoval")
state, control, t0, toff = pkg.synthetic.config2_inputs(traj, B, seed=12345)
other = pkg.synthetic.other_cars(state, seed=777)
knots, V, g = pkg.synthetic.hji_grid_large()
for rho in RHOS:
    mpc = pkg.BatchedTrajectoryTrackingMPC(traj, B, precision="f32", polish_rho=rho)
    mpc.set_hji_cache(knots, V, g)
    mpc.set_inputs(state, control, t0, other_car_state=other, time_offset=toff)
    for _ in range(3):
        mpc.reset(); mpc.step_dev()
    mpc.synchronize()
    report(f"config 3     rho {rho:6.0f} cold      ", mpc)
    mpc.close()
