"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on identical seeded inputs.

Tolerances (fp64 path): intermediates 1e-9 relative-inf (different summation orders / libm vs ocml transcendental functions);
optimal controls 1e-6 relative-inf on NORMALISED controls against the oracle's exact optimum of the SAME QP data (north star);
active-set index lists identical.
"""
import numpy as np
import pytest

from conftest import make_oracle

pytestmark = pytest.mark.gpu

B_SMALL = 192


def rel_inf(a, b, floor=1.0):
    a = np.asarray(a); b = np.asarray(b)
    return float(np.max(np.abs(a - b)) / max(floor, float(np.max(np.abs(b)))))


@pytest.fixture(scope="module")
def setup(pkg, oracle_mod, skidpad):
    mpc = pkg.BatchedTrajectoryTrackingMPC(skidpad, B_SMALL)
    orc = make_oracle(oracle_mod, skidpad)
    state, control, t0, toff = pkg.synthetic.config2_inputs(skidpad, B_SMALL, seed=12345)
    return mpc, orc, state, control, t0, toff


def oracle_pipeline(orc, state, control, t0, toff, b, other=(0, 0, 0, 0)):
    ts, dt = orc.time_steps(t0[b])
    qs, us, ps = orc.nodes(state[b], control[b], ts, dt, time_offset=toff[b])
    sd = orc.update_qp(qs, us, ps, dt, state[b], control[b], other)
    return ts, dt, qs, us, ps, sd


def test_time_steps_and_projection(setup):
    mpc, orc, state, control, t0, toff = setup
    mpc.set_inputs(state, control, t0, time_offset=toff)
    mpc.compute_time_steps_()
    mpc.compute_linearization_nodes_()
    ts, dt, pts = mpc.time_steps()
    sep = mpc.path_coordinates()
    for b in range(B_SMALL):
        ots, odt = orc.time_steps(t0[b])
        assert np.array_equal(ts[b], ots) and np.array_equal(dt[b], odt)
        assert np.array_equal(pts[b], ts[b])          # prev_ts aliases ts in the reference (model_predictive_control.jl:15)
        s, e, t, _ = orc.path_coordinates(state[b, 0], state[b, 1])
        assert abs(sep[b, 0] - s) <= 1e-9 * max(1, abs(s)) and abs(sep[b, 1] - e) <= 1e-9 and abs(sep[b, 2] - t) <= 1e-9 * max(1, abs(t))


def test_time_grid_is_julias_range_arithmetic_bit_for_bit(pkg, oracle_mod, skidpad):
    """model_predictive_control.jl:25-26 build the grid out of Julia ranges (`t0 .+ dt_short*(0:N_short)`, `t0_long .+ dt_long*(1:N_long)`: TwicePrecision reference and step,
    one rounding per element).  Kernel (k_time_steps and the fused time grid of k_project), C++ oracle and the independent Python restatement agree BIT FOR BIT on the knife
    edges of the correction step and on random times; option "time_grid_naive" gives the two-rounding grid of rounds 1-5, which differs.  And the clock of pg_simulate_dev
    (`for t in 0:dt:trajectory.t[end]`, :87) is the range's elements, continued across calls -- not an accumulation.  (A reading of Julia 1.0's Base that could not be
    executed: tests/test_time_grid_ranges.py.)"""
    from oracle import spec_numpy as sp
    rng = np.random.default_rng(3)
    t0 = np.concatenate([[0.0, 0.09, 0.29, 0.49, 0.69, 1.29, 16.09, 12.29, 1e-9, 1234.5678], 0.01 * np.arange(0, 400, 7), rng.uniform(0.0, 150.0, 189)])
    B = len(t0)
    state, control, _, toff = pkg.synthetic.config2_inputs(skidpad, B, seed=5)
    orc = make_oracle(oracle_mod, skidpad)
    mpc = pkg.BatchedTrajectoryTrackingMPC(skidpad, B)
    mpc.set_inputs(state, control, t0, time_offset=toff)
    mpc.compute_time_steps_()
    ts, dt, _ = mpc.time_steps()
    mpc.compute_linearization_nodes_()          # (the step path writes the grid from inside the projection kernel: same bits)
    ts_b, dt_b, _ = mpc.time_steps()
    differ = 0
    for b in range(B):
        ots, odt = orc.time_steps(t0[b]); sts, sdt = sp.compute_time_steps(t0[b])
        assert np.array_equal(ts[b], ots) and np.array_equal(dt[b], odt) and np.array_equal(ots, sts) and np.array_equal(odt, sdt), (b, t0[b])
        differ += not np.array_equal(ots, sp.compute_time_steps(t0[b], naive=True)[0])
    assert np.array_equal(ts, ts_b) and np.array_equal(dt, dt_b) and differ >= 20
    mpc.step_dev(); mpc.synchronize()
    assert np.array_equal(mpc.time_steps()[0], ts)
    nv = pkg.BatchedTrajectoryTrackingMPC(skidpad, B, options={"time_grid_naive": 1})
    nv.set_inputs(state, control, t0, time_offset=toff); nv.compute_time_steps_()
    tsn = nv.time_steps()[0]
    assert all(np.array_equal(tsn[b], sp.compute_time_steps(t0[b], naive=True)[0]) for b in range(B)) and not np.array_equal(tsn, ts)
    # the closed loop's clock: 4 + 40 steps see the times of 44 (k / 100 rounded once for the instance that starts at 0), the host-side helper returns the same numbers
    t_end = float(skidpad.t[-1])
    clock = mpc.simulate_clock(45, t0)
    assert all(np.array_equal(clock[:, b], orc.simulate_times(0.01, t_end, 45, t_start=float(t0[b]))) and np.array_equal(clock[:, b], sp.simulate_times(0.01, t_end, 45, t_start=float(t0[b]))) for b in range(B))
    mpc.set_inputs(state, control, t0, time_offset=toff)
    _, _, t4, _, _ = mpc.simulate_(4)
    assert np.array_equal(t4, clock[4])
    _, _, t44, _, _ = mpc.simulate_(40)
    assert np.array_equal(t44, clock[44]) and t44[0] == sp.simulate_times(0.01, t_end, 45)[44]
    nv.set_inputs(state, control, t0, time_offset=toff)
    _, _, tn, _, _ = nv.simulate_(44)
    acc = t0.copy()
    for _ in range(44):
        acc = acc + 0.01
    assert np.array_equal(tn, acc) and not np.array_equal(tn, t44)
    mpc.close(); nv.close()


def test_cold_nodes(setup):
    mpc, orc, state, control, t0, toff = setup
    mpc.reset()
    mpc.set_inputs(state, control, t0, time_offset=toff)
    mpc.compute_time_steps_(); mpc.compute_linearization_nodes_()
    qs, us, ps = mpc.nodes()
    for b in range(B_SMALL):
        _, _, oq, ou, op, _ = oracle_pipeline(orc, state, control, t0, toff, b)
        assert rel_inf(qs[b], oq) < 1e-9, b
        assert rel_inf(us[b], ou) < 1e-9, b
        assert rel_inf(ps[b], op) < 1e-9, b


def test_qp_data(setup):
    mpc, orc, state, control, t0, toff = setup
    mpc.reset()
    mpc.set_inputs(state, control, t0, time_offset=toff)
    mpc.compute_time_steps_(); mpc.compute_linearization_nodes_(); mpc.update_QP_()
    qp = mpc.qp_data()
    assert qp.shape[1] == orc.sd_len
    for b in range(B_SMALL):
        sd = oracle_pipeline(orc, state, control, t0, toff, b)[5]
        G = orc.unpack_sd(qp[b]); O = orc.unpack_sd(sd)
        for k in O:
            assert rel_inf(G[k], O[k]) < 1e-8, (b, k, rel_inf(G[k], O[k]))


def test_solve_matches_exact_optimum(setup, oracle_mod, pkg):
    mpc, orc, state, control, t0, toff = setup
    mpc.reset()
    u, status, iters = mpc.step_(state, control, t0, time_offset=toff)
    assert np.all(status == pkg.SOLVED), status
    qp = mpc.qp_data()
    x, sg = mpc.solution()
    st, it, act, mu = mpc.solve_info(); lam = mpc.multipliers()
    worst_u2 = 0.0; worst_all = 0.0
    for b in range(B_SMALL):
        xe, ye, info = orc.solve_exact(qp[b])            # exact optimum of the SAME QP data the GPU solved
        assert info["status"] == 1
        X = orc.split_x(xe)
        worst_u2 = max(worst_u2, rel_inf(x[b, 1, 6:], X["u"][1]))
        worst_all = max(worst_all, rel_inf(x[b, :, 6:], X["u"]), rel_inf(x[b, :, :6], X["q"]))
        un = orc.next_control(X["u"][1])
        assert rel_inf(u[b] / [mpc.u_normalization[0], mpc.u_normalization[1], mpc.u_normalization[1]],
                       un / [mpc.u_normalization[0], mpc.u_normalization[1], mpc.u_normalization[1]]) < 1e-6
        qpc = orc.assemble_qp(qp[b])
        assert mpc.canonical_active_set(b, act[b], qp[b], lam=lam[b]) == oracle_mod.active_set(qpc, xe, ye, tol=1e-6), b
    assert worst_u2 < 1e-6, worst_u2
    assert worst_all < 1e-5, worst_all


def test_warm_second_step(setup, pkg):
    """Second consecutive step: warm branch of compute_linearization_nodes! (coupled_lat_long.jl:82-102) on both sides."""
    mpc, orc, state, control, t0, toff = setup
    mpc.reset()
    u1, status, _ = mpc.step_(state, control, t0, time_offset=toff)
    x1, _ = mpc.solution()
    ts1, _, _ = mpc.time_steps()
    # advance the plant with the OLD control (simulate semantics, model_predictive_control.jl:94-95) using the oracle's plant model
    state2 = np.stack([orc.plant_step(state[b], control[b], 0.01) for b in range(B_SMALL)])
    u2, status2, _ = mpc.step_(state2, u1, t0 + 0.01, time_offset=toff)
    assert np.all(status2 == pkg.SOLVED)
    qs, us, ps = mpc.nodes()
    qp = mpc.qp_data()
    x2, _ = mpc.solution()
    for b in range(0, B_SMALL, 4):
        ts, dt = orc.time_steps(t0[b] + 0.01)
        oq, ou, op = orc.nodes(state2[b], u1[b], ts, dt, time_offset=toff[b], solved=True, prev_ts=ts, prev_q=x1[b, :, :6], prev_u=x1[b, :, 6:])
        assert rel_inf(qs[b], oq) < 1e-9 and rel_inf(us[b], ou) < 1e-9 and rel_inf(ps[b], op) < 1e-9
        xe, ye, info = orc.solve_exact(qp[b])
        X = orc.split_x(xe)
        assert rel_inf(x2[b, 1, 6:], X["u"][1]) < 1e-6


def test_gpu_against_the_independent_numpy_spec(pkg):
    """The HIP path against oracle/spec_numpy.py DIRECTLY (not through the C++ oracle): time grid bit-exact, cold nodes 1e-9, refreshed QP data from the same
    nodes 1e-8, and the canonical QP the spec assembles from those data solved to its exact optimum gives the applied control of the GPU (1e-6)."""
    from oracle import spec_numpy as S
    from oracle import oracle as om
    import scipy.sparse as sp
    traj = pkg.load_path_fixture("vail")
    n = 3
    state, control, t0, toff = pkg.synthetic.config2_inputs(traj, n, seed=404, s_range=(5.0, 30.0))
    mpc = pkg.BatchedTrajectoryTrackingMPC(traj, n)
    u, st, _ = mpc.step_(state, control, t0, time_offset=toff)
    assert np.all(st == pkg.SOLVED)
    ts_g, dt_g, _ = mpc.time_steps(); qs_g, us_g, ps_g = mpc.nodes(); qp_g = mpc.qp_data(); x_g, _ = mpc.solution()
    P, U, T = S.X1(), S.coupled_control_params(), S.Trajectory(traj.data)
    for b in range(n):
        ts, dt = S.compute_time_steps(float(t0[b]))
        assert np.array_equal(ts, ts_g[b]) and np.array_equal(dt, dt_g[b])
        qs, us, ps = S.compute_linearization_nodes(P, U, T, state[b], control[b], ts, dt, 10, 20, time_offset=float(toff[b]))
        for mine, theirs in ((qs_g[b], qs), (us_g[b], us), (ps_g[b], ps)):
            assert np.max(np.abs(mine - theirs) / np.maximum(1.0, np.abs(theirs))) < 1e-9, b
        D = S.update_qp(P, U, qs_g[b], us_g[b], ps_g[b], dt, 10, 20)
        assert np.max(np.abs(D["flat"] - qp_g[b]) / np.maximum(1.0, np.abs(D["flat"]))) < 1e-8, b
        Q = S.assemble_canonical_qp(P, U, D, 10, 20)
        A = sp.csc_matrix(Q["A"])
        xe, ye, info = om.solve_exact_generic(dict(Pd=Q["Pd"], q=Q["q"], Ap=A.indptr, Ai=A.indices, Ax=A.data, l=Q["l"], u=Q["u"]))
        assert info["status"] == 1
        u2 = xe[6 * 31 + 2: 6 * 31 + 4]                       # u[:, 2] of the reference's variable order: q (6 x 31), then u (2 x 31)
        assert np.max(np.abs(x_g[b, 1, 6:] - u2)) < 1e-6, b
    mpc.close()
