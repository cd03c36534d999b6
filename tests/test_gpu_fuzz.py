"""GPU stress test: wide random inputs (large tracking errors, heading errors, speed mismatches, arbitrary measured controls, both tracking modes, cold then warm) on
three of the reference's paths.  Whatever the solver reports as SOLVED *and verified* (polish info >= 1) must be the exact optimum of its QP data (oracle, 1e-6);
unverified answers (polish info -1) are rare and within 1e-2; what it gives up on must be a QP
the oracle's exact solver cannot solve either; nothing may come back non-finite unless the status says so."""
import numpy as np
import pytest

from conftest import make_oracle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("path,seed", [("skidpadoval", 1), ("vail", 2), ("EastPaddock", 3)])
def test_wide_random_inputs(pkg, oracle_mod, path, seed):
    traj = pkg.load_path_fixture(path)
    B = 768
    rng = np.random.default_rng(seed)
    s_hi = float(traj.s[-1])
    state, control, t0, toff = pkg.synthetic.config2_inputs(traj, B, seed=seed, traj_mode=True, s_range=(3.0, max(8.0, s_hi - 40.0)))
    # widen: lateral offset up to 2.5 m, heading error up to 0.6 rad, speed 0.6x .. 1.8x, sideslip and yaw-rate disturbances, measured controls anywhere in the actuator range
    psi = state[:, 2].copy()
    e = rng.uniform(-2.5, 2.5, B)
    state[:, 0] -= e * np.cos(psi); state[:, 1] -= e * np.sin(psi)
    state[:, 2] += rng.uniform(-0.6, 0.6, B)
    state[:, 3] = np.clip(state[:, 3] * rng.uniform(0.6, 1.8, B), 1.2, 14.5)
    state[:, 4] = rng.uniform(-1.0, 1.0, B); state[:, 5] += rng.uniform(-0.5, 0.5, B)
    X = pkg.X1()
    d0 = rng.uniform(-0.95, 0.95, B) * X["delta_max"]; Fx0 = rng.uniform(0.95 * X["Fx_min"], 0.95 * X["Fx_max"], B)
    control = np.stack([d0, np.where(Fx0 > 0, 0.0, 0.6) * Fx0, np.where(Fx0 > 0, 1.0, 0.4) * Fx0], axis=1)
    toff = np.where(rng.uniform(size=B) < 0.5, 0.0, np.nan)
    mpc = pkg.BatchedTrajectoryTrackingMPC(traj, B)
    orc = make_oracle(oracle_mod, traj)
    un = np.array([mpc.u_normalization[0], mpc.u_normalization[1], mpc.u_normalization[1]])
    for step in range(2):                                   # cold step, then a warm step from the plant-advanced state (warm start of the active set)
        u, st, it = mpc.step_(state, control, t0, time_offset=toff)
        qp = mpc.qp_data(); x, _ = mpc.solution(); pol = mpc.polish_info()
        ok = pkg.is_solved(st)
        # status honesty: PG_SOLVED is returned for VERIFIED KKT points only; an interior-point iterate no polish round verified is PG_SOLVED_UNVERIFIED
        assert np.array_equal(st[ok] == pkg.SOLVED, pol[ok] >= 1)
        assert np.mean(ok) > 0.7, (path, step, np.bincount(st))
        assert np.all(np.isfinite(u[ok])) and np.all(np.isfinite(x[ok]))
        # PG_NUMERICAL with finite inputs is legitimate in exactly one situation, and the reference shares it: the explicit RK4 of `linearize` is unstable at low
        # speed (|lambda h| > 2.78 below ~2 m/s) and overflows when the warm nodes come from a wild previous solution -- the QP data are then 1e40+
        num = st == pkg.NUMERICAL
        assert np.all((st == pkg.SOLVED) | (st == pkg.SOLVED_UNVERIFIED) | (st == pkg.MAX_ITER) | (st == pkg.INFEASIBLE_X0) | num), np.bincount(st)
        assert num.sum() <= 4 and all(not np.all(np.isfinite(qp[b])) or np.max(np.abs(qp[b])) > 1e10 for b in np.flatnonzero(num)), (path, step, int(num.sum()))
        worst, worst_unverified, n_bad = 0.0, 0.0, 0
        idx = rng.choice(B, 192, replace=False)
        for b in idx:
            if ok[b]:
                xe, ye, info = orc.solve_exact(qp[b])
                if info["status"] != 1:
                    continue                                 # (an instance the oracle's own solver gives up on proves nothing either way)
                err = float(np.max(np.abs(x[b, 1, 6:] - orc.split_x(xe)["u"][1])))
                if pol[b] >= 1: worst = max(worst, err)
                else: worst_unverified = max(worst_unverified, err)
        # a VERIFIED point (pg_get_polish_info >= 1) is a KKT point of the QP: exact.  Where neither polish attempt verifies (-1: 1-3 instances in 4608 in this regime,
        # none in normal tracking) the answer is the interior-point iterate at its rounding floor -- still a good control, not an exact one, and flagged as such
        assert worst < 1e-6, (path, step, worst)
        assert worst_unverified < 1e-2 and np.mean(pol[ok] < 1) < 0.02, (path, step, worst_unverified, float(np.mean(pol[ok] < 1)))
        for b in [b for b in idx if st[b] == pkg.MAX_ITER][:4]:
            xe, ye, info = orc.solve_exact(qp[b])
            n_bad += int(info["status"] == 1 and info["iters"] >= 0)      # (iters < 0: the oracle's own interior point gave up too and its ADMM fall-back answered, after 1e3-1e5 iterations)
        assert n_bad == 0, (path, step, "gave up on a QP the oracle solves")
        if step == 1:
            assert np.mean(it[ok] == 0) > 0.15           # a share of the warm instances is served by the warm polish alone even in this regime (30-50 %; > 99 % in normal tracking)
        state = np.stack([orc.plant_step(state[b], control[b], 0.01) for b in range(B)]); control = np.where(ok[:, None], u, control); t0 = t0 + 0.01
        if not np.all(ok):
            mpc.reset(mask=~ok)                         # what the ROS loop does with a controller that did not deliver (ros_integration.jl:134-147: solved = false)
    mpc.close()
