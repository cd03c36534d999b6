"""GPU parity tests of the HJI lookup / safety-constraint path (HJI_computation.jl:20-24,66-131,160-170) against the CPU oracle."""
import math

import numpy as np
import pytest

from conftest import make_oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def grid(pkg):
    return pkg.synthetic.hji_grid(dims=(7, 6, 5, 4, 4, 5, 4), seed=11)


@pytest.mark.parametrize("cell_dims", [7, 5, 3])
def test_lookup_matches_oracle(pkg, oracle_mod, skidpad, grid, cell_dims, monkeypatch):
    """All three device layouts (256 B cell records = the default, 1 KiB, 4 KiB: option "hji_cell_dims") against the oracle."""
    knots, V, g = grid
    mpc = pkg.BatchedTrajectoryTrackingMPC(skidpad, 8, options={"hji_cell_dims": cell_dims})
    mpc.set_hji_cache(knots, V, g)
    orc = make_oracle(oracle_mod, skidpad); orc.set_hji_grid(knots, V, g)
    rng = np.random.default_rng(0)
    n = 500
    lo = np.array([k[0] for k in knots], dtype=np.float64); hi = np.array([k[-1] for k in knots], dtype=np.float64)
    x = lo + (hi - lo) * rng.uniform(-0.05, 1.05, (n, 7))          # ~30 % of the points fall outside the grid
    x[0] = lo; x[1] = hi                                            # exact corners are in bounds (:67 uses <=)
    x[2] = [float(k[len(k) // 2]) for k in knots]                   # an interior knot point
    Vg, Gg = mpc.hji_lookup(x)
    nin = 0
    for i in range(n):
        Vo, Go, inb = orc.hji_lookup(x[i])
        if inb:
            nin += 1
            assert abs(Vg[i] - Vo) <= 1e-12 * max(1.0, abs(Vo)), i
            assert np.max(np.abs(Gg[i] - Go)) <= 1e-12, i
        else:
            assert math.isinf(Vg[i]) and Vg[i] > 0 and np.all(Gg[i] == 0), i
    assert 50 < nin < n


def test_constraint_rows_and_solve_with_hji(pkg, oracle_mod, skidpad, grid):
    """update_QP! with an active safety row (V <= eps), then the full solve against the exact optimum of the same QP."""
    knots, V, g = grid
    B = 48
    mpc = pkg.BatchedTrajectoryTrackingMPC(skidpad, B, hji_eps=10.0)            # large eps so that most rows are active
    mpc.set_hji_cache(knots, V, g)
    orc = make_oracle(oracle_mod, skidpad); orc.set_hji_grid(knots, V, g); orc.set_hji_eps(10.0)
    state, control, t0, toff = pkg.synthetic.config2_inputs(skidpad, B, seed=21)
    other = pkg.synthetic.other_cars(state, seed=5)
    u, status, iters = mpc.step_(state, control, t0, other_car_state=other, time_offset=toff)
    M, b, Vv = mpc.hji_constraint()
    nact = 0
    for i in range(B):
        Mo, bo, Vo = orc.hji_constraint(state[i], other[i], control[i])
        Mo = Mo * orc.u_norm                                                     # coupled_lat_long.jl:345
        if math.isinf(Vo):
            assert math.isinf(Vv[i]) and np.all(M[i] == 0) and b[i] == 1.0
        else:
            nact += 1
            assert abs(Vv[i] - Vo) <= 1e-12 * max(1, abs(Vo))
            assert np.max(np.abs(M[i] - Mo)) <= 1e-9 * max(1.0, np.max(np.abs(Mo))) and abs(b[i] - bo) <= 1e-9 * max(1.0, abs(bo)), i
    assert nact >= B // 2
    assert np.all(status == pkg.SOLVED), status
    qp = mpc.qp_data(); x, sg = mpc.solution(); st, it, act, mu = mpc.solve_info(); lam = mpc.multipliers()
    worst = 0.0
    for i in range(B):
        xe, ye, info = orc.solve_exact(qp[i]); X = orc.split_x(xe)
        worst = max(worst, float(np.max(np.abs(x[i, 1, 6:] - X["u"][1]))))
        qpc = orc.assemble_qp(qp[i])
        assert mpc.canonical_active_set(i, act[i], qp[i], lam=lam[i]) == oracle_mod.active_set(qpc, xe, ye, tol=1e-6), i
        # the penalised slacks of nodes 2 and 3 (sigma_HJI, N_HJI = 3) match the canonical solution
        assert np.max(np.abs(sg[i, :2, 2] - X["sigma_hji"][1:3])) < 1e-6
    assert worst < 1e-6, worst


def test_nan_hazard_is_reported(pkg, skidpad, grid):
    """other-car speed 0 inside the grid with V <= eps gives b_HJI = NaN in the reference; the kernel must flag it."""
    knots, V, g = grid
    mpc = pkg.BatchedTrajectoryTrackingMPC(skidpad, 4, hji_eps=10.0)
    mpc.set_hji_cache(knots, V, g)
    state, control, t0, toff = pkg.synthetic.config2_inputs(skidpad, 4, seed=3)
    other = pkg.synthetic.other_cars(state, seed=5)
    other[1, 3] = 0.0
    u, status, iters = mpc.step_(state, control, t0, other_car_state=other, time_offset=toff)
    assert status[1] == pkg.NUMERICAL and np.all(np.isnan(u[1]))
    assert status[0] == pkg.SOLVED and status[2] == pkg.SOLVED and np.all(np.isfinite(u[[0, 2, 3]]))


def test_hji_fallback_policy(pkg, oracle_mod, skidpad, grid):
    """optimal_control (HJI_computation.jl:133-158) and the ROS loop's selection (ros_integration.jl:114-124) against the oracle:
    bang-bang steer sign and the argmax of the 50-point Fx line search are index work (exact), the chosen forces are bit-identical grid points."""
    knots, V, g = grid
    B = 96
    eps = 0.5
    mpc = pkg.BatchedTrajectoryTrackingMPC(skidpad, B, hji_eps=eps)
    mpc.set_hji_cache(knots, V, g)
    orc = make_oracle(oracle_mod, skidpad); orc.set_hji_grid(knots, V, g); orc.set_hji_eps(eps)
    state, control, t0, toff = pkg.synthetic.config2_inputs(skidpad, B, seed=33)
    toff[::5] = np.nan                                               # path-tracking mode never hands over (tracking_mode != :traj)
    other = pkg.synthetic.other_cars(state, seed=9)
    u_mpc, status, _ = mpc.step_(state, control, t0, other_car_state=other, time_offset=toff)
    assert np.all(status == pkg.SOLVED)
    u_on, src_on, u2 = mpc.get_next_control_hji(True)
    u_off, src_off, _ = mpc.get_next_control_hji(False)
    P = mpc.vehicle
    fx_grid = np.array([n / 49 * P["Fx_max"] + (1 - n / 49) * P["Fx_min"] for n in range(50)])
    seen = set()
    for i in range(B):
        Vo, u2o = orc.hji_optimal_control(state[i], other[i])
        unsafe = (not math.isnan(toff[i])) and Vo <= eps
        assert src_on[i] == (1 if unsafe else 0) and src_off[i] == (2 if unsafe else 0), i
        seen.add(int(src_on[i]))
        assert np.array_equal(u_off[i], u_mpc[i])
        if not math.isinf(Vo):
            assert u2[i, 0] == u2o[0], i                                                         # steer sign: exact
            assert int(np.argmin(np.abs(fx_grid - u2[i, 1]))) == int(np.argmin(np.abs(fx_grid - u2o[1]))), i   # line-search index: exact
            assert u2[i, 1] == u2o[1], i
        if unsafe:
            Fx = u2o[1]
            exp = [u2o[0], Fx * (P["fwd_frac"] if Fx > 0 else P["fwb_frac"]), Fx * (P["rwd_frac"] if Fx > 0 else P["rwb_frac"])]
            assert np.allclose(u_on[i], exp, rtol=0, atol=1e-12 * max(1.0, abs(Fx))), i
        else:
            assert np.array_equal(u_on[i], u_mpc[i])
    assert seen == {0, 1}, seen                                      # both outcomes exercised
    mpc.close()


@pytest.mark.parametrize("precision", ["f64", "f32"])
def test_value_slices_colours_and_zero_contour(pkg, oracle_mod, skidpad, precision):
    """SURVEY 8(f) N4, the RViz consumers (rviz.jl:23-40 values marker, :60-69 contour marker) as one batched call: V at every (x, y) knot pair, the marker
    colours, and the zero-level crossings, against the oracle's restatement; the traced line is a walk along V = 0."""
    knots, V, g = pkg.synthetic.hji_grid(dims=(21, 17, 5, 4, 4, 5, 4), seed=4)
    mpc = pkg.BatchedTrajectoryTrackingMPC(skidpad, 8, precision=precision)
    mpc.set_hji_cache(knots, V, g)
    orc = make_oracle(oracle_mod, skidpad); orc.set_hji_grid(knots, V, g)
    rng = np.random.default_rng(1)
    lo = np.array([k[0] for k in knots], dtype=np.float64); hi = np.array([k[-1] for k in knots], dtype=np.float64)
    q = lo + (hi - lo) * rng.uniform(0.1, 0.9, (6, 7))
    q[5, 5] = hi[5] + 1.0                                            # one relative state outside the grid: the whole slice is +Inf, no contour
    if precision == "f32":
        q = q.astype(np.float32).astype(np.float64)
    Vs, rgb, cx, cy = mpc.hji_value_slice(q, colors=True)
    tol = 1e-12 if precision == "f64" else 2e-6
    n_lines = 0
    for b in range(6):
        Vo, rgbo, cxo, cyo = orc.hji_slice(knots, q[b])
        if b == 5:
            assert np.all(np.isinf(Vs[b])) and np.all(np.isnan(cx[b])) and np.all(np.isnan(cy[b]))
            continue
        assert np.max(np.abs(Vs[b] - Vo)) <= tol * max(1.0, np.max(np.abs(Vo)))
        assert np.max(np.abs(rgb[b] - rgbo)) <= max(tol, 1e-12) * 10
        if precision == "f64":                                       # (in fp32 a value within rounding of 0 may change sides)
            assert np.array_equal(np.isnan(cx[b]), np.isnan(cxo)) and np.array_equal(np.isnan(cy[b]), np.isnan(cyo))
            assert np.nanmax(np.abs(cx[b] - cxo), initial=0.0) <= 1e-10 and np.nanmax(np.abs(cy[b] - cyo), initial=0.0) <= 1e-10
        line = pkg.trace_zero_contour(knots[0], knots[1], cx[b], cy[b])
        if line:
            n_lines += 1
            for (x, y) in line[:: max(1, len(line) // 8)]:          # V vanishes at the traced vertices (bilinear interpolation along a grid edge is exact)
                assert abs(mpc.hji_lookup(np.array([x, y] + list(q[b, 2:])))[0][0]) <= (1e-9 if precision == "f64" else 1e-4)
    assert n_lines >= 3
    mpc.close()


def test_pipelined_launch_with_the_safety_row(pkg, skidpad):
    """pg_set_pipeline with an HJI grid installed: (M, b) are computed before the pipelined nodes + update_QP launch and the launch order is re-filed after it.
    Same kernels on the same inputs.  fp64: the safety rows, the QP data and the controls are bit-identical to the launch-per-phase sequence.  fp32: the rows are
    bit-identical, the QP data agree to fp32 rounding (the compiler contracts the fp32 tangent arithmetic differently inside the larger kernel), the controls to the
    sensitivity of an fp32 solve to such rounding."""
    knots, V, g = pkg.synthetic.hji_grid(dims=(7, 6, 5, 4, 4, 5, 4), seed=11)
    n = 2048
    state, control, t0, toff = pkg.synthetic.config2_inputs(skidpad, n, seed=31)
    other = pkg.synthetic.other_cars(state, seed=6)
    for prec in ("f64", "f32"):
        out = {}
        for piped in (False, True):
            mpc = pkg.BatchedTrajectoryTrackingMPC(skidpad, n, precision=prec, hji_eps=10.0)
            mpc.set_hji_cache(knots, V, g)
            mpc.set_pipeline(piped)
            mpc.set_inputs(state, control, t0, other_car_state=other, time_offset=toff)
            mpc.step_dev(); mpc.synchronize()
            M, b, Vv = mpc.hji_constraint()
            un2 = np.asarray(mpc.u_normalization, dtype=float)
            out[piped] = [M.copy(), b.copy(), mpc.qp_data().copy(), mpc.get_next_control().copy(), mpc.solve_info()[0].copy(), mpc.solve_info()[1].copy()]
            mpc.close()
        assert np.sum(out[True][0] != 0.0) > 50                                   # rows are active
        assert np.mean(out[True][4] == pkg.SOLVED) > 0.9
        if prec == "f64":
            for a, c in zip(out[False], out[True]):
                assert np.array_equal(a, c, equal_nan=True)
        else:
            assert np.array_equal(out[False][0], out[True][0]) and np.array_equal(out[False][1], out[True][1])
            qa, qc = out[False][2], out[True][2]
            assert np.max(np.abs(qa - qc) / np.maximum(1.0, np.abs(qa))) < 5e-6
            both = (out[False][4] == pkg.SOLVED) & (out[True][4] == pkg.SOLVED)
            un = np.array([un2[0], un2[1], un2[1]])
            err = np.abs(out[False][3][both] - out[True][3][both]) / un
            assert np.median(err) < 1e-4 and err.max() < 2e-2, (np.median(err), err.max())
