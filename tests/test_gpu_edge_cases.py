"""GPU parity on the branches the synthetic benchmark inputs never reach: non-finite poses, saturated actuators (zero B columns,
vehicle_dynamics.jl:293-298), sliding tires (:43-47), friction-saturated seeding (:331-338), arclength past the path end (trajectories.jl:59), a first
node outside the hard bounds, the iteration cap, and the second interior-point start."""
import numpy as np
import pytest

from conftest import make_oracle

pytestmark = pytest.mark.gpu


def oracle_qp(orc, state, control, t0, toff):
    ts, dt = orc.time_steps(t0)
    qs, us, ps = orc.nodes(state, control, ts, dt, time_offset=toff)
    return qs, us, ps, orc.update_qp(qs, us, ps, dt, state, control)


def check_against_oracle(pkg, mpc, orc, state, control, t0, toff, skip=(), node_tol=1e-9):
    """Stage by stage against the oracle for every instance not listed in `skip`: linearization nodes (node_tol), update_QP! from the SAME nodes (1e-8),
    applied control against the exact optimum of the SAME QP data (1e-6, solved instances).  Returns the statuses and the QP data."""
    u, st, it = mpc.step_(state, control, t0, time_offset=toff)
    qp = mpc.qp_data(); x, _ = mpc.solution(); qsg, usg, psg = mpc.nodes()
    worst = 0.0
    for b in range(len(t0)):
        if b in skip:
            continue
        ts, dt = orc.time_steps(t0[b])
        qs, us, ps = orc.nodes(state[b], control[b], ts, dt, time_offset=toff[b])
        for mine, theirs in ((qsg[b], qs), (usg[b] / orc.u_norm, us / orc.u_norm), (psg[b], ps)):
            assert np.max(np.abs(mine - theirs) / np.maximum(1.0, np.abs(theirs))) < node_tol, b
        sd = orc.update_qp(qsg[b], usg[b], psg[b], dt, state[b], control[b])
        assert np.max(np.abs(sd - qp[b]) / np.maximum(1.0, np.abs(sd))) < 1e-8, b
        if st[b] == pkg.SOLVED:
            xe, ye, info = orc.solve_exact(qp[b])
            assert info["status"] == 1, b
            worst = max(worst, float(np.max(np.abs(x[b, 1, 6:] - orc.split_x(xe)["u"][1]))))
    assert worst < 1e-6, worst
    return st, qp


def test_non_finite_pose_does_not_fault_and_does_not_spread(pkg, skidpad):
    """a NaN / Inf position made k_project index the trajectory with its sentinel.  Now the instance is poisoned (PG_NUMERICAL) and its
    neighbours are bit-identical to a batch without it."""
    B = 256
    state, control, t0, toff = pkg.synthetic.config2_inputs(skidpad, B, seed=77)
    bad = state.copy()
    bad[3, 0] = np.nan; bad[100, 1] = np.inf; bad[200, 0] = -np.inf; bad[255, 2] = np.nan
    mpc = pkg.BatchedTrajectoryTrackingMPC(skidpad, B)
    u0, st0, _ = mpc.step_(state, control, t0, time_offset=toff)
    mpc.reset()
    u1, st1, _ = mpc.step_(bad, control, t0, time_offset=toff)
    hit = np.zeros(B, bool); hit[[3, 100, 200, 255]] = True
    assert np.all(st1[hit] == pkg.NUMERICAL), st1[hit]
    assert np.all(st1[~hit] == pkg.SOLVED) and np.array_equal(u1[~hit], u0[~hit])
    mpc.close()


def test_saturated_actuators_and_sliding_tires(pkg, skidpad, oracle_mod):
    """Measured control AT / BEYOND the actuator limits (clamp active at the first linearization node => zero columns in B, ForwardDiff semantics of
    apply_control_limits) and large slip (Fiala sliding branch), still feasible for the rate limits."""
    B = 96
    state, control, t0, toff = pkg.synthetic.config2_inputs(skidpad, B, seed=5)
    rng = np.random.default_rng(6)
    X = pkg.X1()
    dmax = X["delta_max"]
    sgn = np.where(rng.uniform(size=B) < 0.5, -1.0, 1.0)
    delta0 = sgn * (dmax + rng.uniform(0.0, 0.003, B))                      # beyond delta_max, within one step of the rate limit (0.344 rad/s x 10 ms)
    Fx0 = np.where(rng.uniform(size=B) < 0.5, rng.uniform(5600, 9000, B), rng.uniform(0.999 * X["Fx_min"], 0.9 * X["Fx_min"], B))      # above Fx_max / close to Fx_min
    control = np.stack([delta0, np.where(Fx0 > 0, 0.0, 0.6) * Fx0, np.where(Fx0 > 0, 1.0, 0.4) * Fx0], axis=1)
    state[:48, 4] = rng.uniform(-3.0, 3.0, 48)                               # Uy up to half of Ux: slip angles far past the sliding threshold
    state[:48, 5] += rng.uniform(-0.8, 0.8, 48)
    mpc = pkg.BatchedTrajectoryTrackingMPC(skidpad, B)
    orc = make_oracle(oracle_mod, skidpad)
    st, qp = check_against_oracle(pkg, mpc, orc, state, control, t0, toff)
    N = mpc.N
    B0 = qp[:, 36 * N:48 * N].reshape(B, N, 6, 2)
    assert np.all(B0[:, 0, :, 0] == 0.0)                                    # d/d(delta) at node 1: the clamp is active for every instance
    assert np.all(B0[Fx0 > 5600][:, 0, :, 1] == 0.0)                        # d/d(Fx) where Fx sits above Fx_max
    assert np.mean(st == pkg.SOLVED) > 0.5, np.bincount(st)
    mpc.close()


def test_friction_saturated_seeding(pkg, oracle_mod):
    """Two to four times the path speed on the tight `vail` loop (V^2 kappa far above mu g): steady_state_estimates runs its friction-limited branches
    (vehicle_dynamics.jl:331-338).  The reference's inverse tire model is discontinuous exactly where those branches put the front tire
    (tests/test_spec_numpy.py::test_inverse_tire_model_is_discontinuous_at_saturation): on such a node the seeded delta is decided by the last bit in ANY
    implementation, so instances whose chain touches the jump (front force within 1e-9 of the friction circle, read off the numpy spec) are excluded."""
    from oracle import spec_numpy as S
    traj = pkg.load_path_fixture("vail")
    B = 192
    state, control, t0, toff = pkg.synthetic.config2_inputs(traj, B, seed=15, s_range=(5.0, float(traj.s[-1]) - 80.0))
    state[:, 3] *= np.linspace(1.5, 4.5, B)
    orc = make_oracle(oracle_mod, traj)
    P, U, T = S.X1(), S.coupled_control_params(), S.Trajectory(traj.data)
    on_jump, saturated = set(), 0
    for b in range(B):
        S.SATURATION_MARGINS = []
        ts, dt = S.compute_time_steps(float(t0[b]))
        qs, us, ps = S.compute_linearization_nodes(P, U, T, state[b], control[b], ts, dt, 10, 20, time_offset=float(toff[b]))
        if min(S.SATURATION_MARGINS) < 1e-9:
            on_jump.add(b)
        saturated += int(np.max(qs[11:, 1] ** 2 * np.abs(ps[11:, 1])) > 0.8 * P["mu"] * P["G"])
    S.SATURATION_MARGINS = None
    assert saturated > B // 8 and len(on_jump) < B - 24, (saturated, len(on_jump))
    mpc = pkg.BatchedTrajectoryTrackingMPC(traj, B)
    # the friction-limited branch takes sqrt(A_max^2 - A_rad^2) (:336) arbitrarily close to zero: rounding differences of 1e-16 come out as 1e-8 and the
    # 30-node recurrence carries them on, so the seeds are compared at 1e-4 here (update_QP! and the solve are still checked at full accuracy)
    st, _ = check_against_oracle(pkg, mpc, orc, state, control, t0, toff, skip=on_jump, node_tol=1e-4)
    mpc.close()


def test_horizon_runs_past_the_end_of_the_path(pkg, oracle_mod):
    """Start 3..12 m before the last path node: most of the 4.1 s horizon lies beyond s_end (`s > traj.s[end]` branch of traj[s], trajectories.jl:59, and
    the Line() extrapolation of interp_by_s, :32-35)."""
    traj = pkg.load_path_fixture("EastPaddock")
    B = 64
    s_end = float(traj.s[-1])
    state, control, t0, toff = pkg.synthetic.config2_inputs(traj, B, seed=3, s_range=(s_end - 12.0, s_end - 3.0))
    mpc = pkg.BatchedTrajectoryTrackingMPC(traj, B)
    orc = make_oracle(oracle_mod, traj)
    st, _ = check_against_oracle(pkg, mpc, orc, state, control, t0, toff)
    assert np.all(st == pkg.SOLVED)
    mpc.close()


def test_first_node_outside_the_hard_bounds(pkg, skidpad):
    """Ux_1 outside [V_min, V_max] or Fx_1 < Fx_min: rows C5-C7 of the reference QP are violated by the FIXED first node (coupled_lat_long.jl:244-251):
    the QP is infeasible whatever the solver does; reported as PG_INFEASIBLE_X0 for exactly those instances."""
    B = 64
    state, control, t0, toff = pkg.synthetic.config2_inputs(skidpad, B, seed=9)
    X = pkg.X1()
    state[0, 3] = 0.5; state[1, 3] = 16.0
    control[2] = [0.0, 0.6 * 1.05 * X["Fx_min"], 0.4 * 1.05 * X["Fx_min"]]
    mpc = pkg.BatchedTrajectoryTrackingMPC(skidpad, B)
    u, st, it = mpc.step_(state, control, t0, time_offset=toff)
    assert np.all(st[:3] != pkg.SOLVED) and pkg.INFEASIBLE_X0 in st[:3]
    assert np.all(st[3:] == pkg.SOLVED)
    mpc.close()


def test_iteration_cap_is_reported(pkg, skidpad):
    B = 64
    state, control, t0, toff = pkg.synthetic.config2_inputs(skidpad, B, seed=10)
    mpc = pkg.BatchedTrajectoryTrackingMPC(skidpad, B, ipm_max_iter=1, cold_guess=0)
    u, st, it = mpc.step_(state, control, t0, time_offset=toff)
    assert np.all(st == pkg.MAX_ITER) and np.all(np.isfinite(u))             # the iterate at the cap is still a finite, dynamics-feasible point
    mpc.close()
    # with the active-set guess on (the default) the cap only binds for the instances the guess does not serve: those come back MAX_ITER, the others SOLVED with
    # zero interior-point iterations
    mpc = pkg.BatchedTrajectoryTrackingMPC(skidpad, B, ipm_max_iter=1)
    u, st, it = mpc.step_(state, control, t0, time_offset=toff)
    pol = mpc.polish_info()
    assert np.all((st == pkg.SOLVED) == (it == 0)) and np.all(pol[st == pkg.SOLVED] >= 1) and np.all(st[it > 0] == pkg.MAX_ITER) and np.mean(it == 0) > 0.5
    mpc.close()


def test_second_start_is_reached_and_lands_on_the_same_optimum(pkg, skidpad, oracle_mod):
    """A cap of 7 iterations stops the first start (v = 0 roll-out) short for part of the batch; the second attempt (least-squares start, 3x the cap: it
    needs 13-24 iterations) must then deliver the same optimum (k_solve, `attempt == 1`)."""
    B = 128
    state, control, t0, toff = pkg.synthetic.config2_inputs(skidpad, B, seed=12)
    ref = pkg.BatchedTrajectoryTrackingMPC(skidpad, B)
    u0, st0, it0 = ref.step_(state, control, t0, time_offset=toff)
    mpc = pkg.BatchedTrajectoryTrackingMPC(skidpad, B, ipm_max_iter=7, cold_guess=0)      # (the active-set guess off: every instance goes through the interior point)
    u1, st1, it1 = mpc.step_(state, control, t0, time_offset=toff)
    second = it1 > 7
    assert second.sum() >= B // 8, int(second.sum())
    ok = st1 == pkg.SOLVED
    assert np.mean(ok[second]) > 0.9, np.bincount(st1[second])
    un = np.array([ref.u_normalization[0], ref.u_normalization[1], ref.u_normalization[1]])
    assert np.max(np.abs(u1[ok] - u0[ok]) / un) < 1e-6
    ref.close(); mpc.close()


def test_distance_to_the_reference_solver_at_its_own_tolerance(pkg, skidpad, oracle_mod):
    """What "drop-in" changes for a maintainer: the reference stops OSQP at eps_abs = eps_rel = 1e-3 (coupled_lat_long.jl:201-203 leaves the defaults), this
    library returns the exact optimum.  Against the oracle's OSQP port (same algorithm and settings as the reference) on the same QP data the applied
    control differs by at most a few 1e-2 (normalised) -- inside what OSQP's own termination rule allows: its iterate violates the constraints by up to
    eps_abs + eps_rel |Ax| and the exact optimum sits within that band.  The exact optimum is never worse in objective than OSQP's point projected back."""
    B = 64
    state, control, t0, toff = pkg.synthetic.config2_inputs(skidpad, B, seed=20)
    mpc = pkg.BatchedTrajectoryTrackingMPC(skidpad, B)
    u, st, _ = mpc.step_(state, control, t0, time_offset=toff)
    qp = mpc.qp_data(); x, _ = mpc.solution()
    orc = make_oracle(oracle_mod, skidpad)
    d_u2, viol = [], []
    for b in range(B):
        orc.reset_instance(0)
        xo, yo, info = orc.osqp_solve(qp[b], inst=0)
        assert info["status"] == 1
        X = orc.split_x(xo)
        d_u2.append(np.max(np.abs(x[b, 1, 6:] - X["u"][1])))
        viol.append(info["res_pri"])
    d_u2 = np.array(d_u2)
    assert d_u2.max() <= 0.1 and np.median(d_u2) <= 2e-2, (d_u2.max(), np.median(d_u2))
    assert d_u2.max() >= 1e-6                   # ... and the difference is real: OSQP at 1e-3 is NOT the optimum (so parity is stated against the optimum)
    assert max(viol) <= 1e-3 * 20               # OSQP's own primal residual at termination (eps_abs + eps_rel * |Ax|_inf)
    mpc.close()
