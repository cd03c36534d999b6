"""CPU tests: the two independent restatements of the reference (oracle/*.hpp in C++, oracle/spec_numpy.py in numpy, written separately from the Julia
files) agree on the whole construction path -- time grid, path projection, linearization nodes (cold and warm), `linearize`, refreshed QP data, and the
canonical QP (row by row).  SURVEY.md section 4 "test pyramid" / section 7.1 step 1."""
import numpy as np
import pytest
import scipy.sparse as sp

from conftest import make_oracle
from oracle import spec_numpy as S

NS, NL = 10, 20
CASES = [("skidpadoval", 0, True), ("skidpadoval", 3, False), ("vail", 1, True), ("EastPaddock", 2, True), ("variable_speed", 4, False)]


def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))))


@pytest.fixture(scope="module")
def ctx(pkg, oracle_mod):
    out = {}
    for path in sorted({c[0] for c in CASES}):
        traj = pkg.load_path_fixture(path)
        s_hi = float(traj.s[-1])
        st, ct, t0, toff = pkg.synthetic.config2_inputs(traj, 8, seed=31, s_range=(5.0, max(10.0, s_hi - 60.0)))
        out[path] = (traj, make_oracle(oracle_mod, traj), S.Trajectory(traj.data), st, ct, t0)
    return out


def test_vehicle_and_params_match(pkg):
    X = pkg.X1(); P = S.X1()
    for k, v in P.items():
        assert X[k] == pytest.approx(v, rel=1e-15), k
    U = pkg.CoupledControlParams(); V = S.coupled_control_params()
    for k, v in V.items():
        assert U[k] == pytest.approx(v, rel=1e-15), k


@pytest.mark.parametrize("path,k,traj_mode", CASES)
def test_construction_path_agrees(ctx, path, k, traj_mode):
    traj, orc, T, st, ct, t0 = ctx[path]
    P, U = S.X1(), S.coupled_control_params()
    state, control, t = st[k], ct[k], float(t0[k])
    toff = 0.0 if traj_mode else float("nan")
    # T1
    ts_o, dt_o = orc.time_steps(t)
    ts_s, dt_s = S.compute_time_steps(t)
    assert np.array_equal(ts_o, ts_s) and np.array_equal(dt_o, dt_s)
    # P1
    s_o, e_o, t_o, _ = orc.path_coordinates(state[0], state[1])
    s_s, e_s, t_s = T.path_coordinates(state[0], state[1])
    assert abs(s_o - s_s) < 1e-9 and abs(e_o - e_s) < 1e-9 and abs(t_o - t_s) < 1e-9      # sqrt(w.w - d2) cancels: 1e-9 is what fp64 leaves of it
    # N1/N3 cold nodes
    q_o, u_o, p_o = orc.nodes(state, control, ts_o, dt_o, time_offset=toff)
    q_s, u_s, p_s = S.compute_linearization_nodes(P, U, T, state, control, ts_s, dt_s, NS, NL, time_offset=toff)
    assert rel(q_s, q_o) < 1e-9 and rel(u_s, u_o) < 1e-9 and rel(p_s, p_o) < 1e-9
    # L1/L2 + Q2: refreshed QP data from the SAME nodes (so that the comparison is 1e-10, not limited by the projection's cancellation)
    sd_o = orc.update_qp(q_o, u_o, p_o, dt_o, state, control)
    D = S.update_qp(P, U, q_o, u_o, p_o, dt_o, NS, NL)
    assert D["flat"].shape == sd_o.shape
    assert rel(D["flat"], sd_o) < 1e-10
    # Q1: canonical QP, row by row (a row may be stated with the opposite sign: then its bounds swap)
    Qo = orc.assemble_qp(sd_o)
    Ao = sp.csc_matrix((Qo["Ax"], Qo["Ai"], Qo["Ap"]), shape=(len(Qo["l"]), len(Qo["Pd"]))).toarray()
    Qs = S.assemble_canonical_qp(P, U, D, NS, NL)
    assert Qs["A"].shape == Ao.shape == (691, 378)
    assert rel(Qs["Pd"], Qo["Pd"]) < 1e-12 and rel(Qs["q"], Qo["q"]) < 1e-12
    big = lambda v: np.where(np.abs(v) > 1e19, np.sign(v) * np.inf, v)
    for i in range(Ao.shape[0]):
        same = rel(Qs["A"][i], Ao[i]) < 1e-10 and rel(big(Qs["l"][i]), big(Qo["l"][i])) < 1e-10 if np.isfinite(big(Qo["l"][i])) else rel(Qs["A"][i], Ao[i]) < 1e-10 and big(Qs["l"][i]) == big(Qo["l"][i])
        same = same and (rel(big(Qs["u"][i]), big(Qo["u"][i])) < 1e-10 if np.isfinite(big(Qo["u"][i])) else big(Qs["u"][i]) == big(Qo["u"][i]))
        if not same:      # negated statement of the same row
            assert rel(-Qs["A"][i], Ao[i]) < 1e-10, i
            lo, hi = -big(Qs["u"][i]), -big(Qs["l"][i])
            for a, b in ((lo, big(Qo["l"][i])), (hi, big(Qo["u"][i]))):
                assert (a == b) if not np.isfinite(b) else abs(a - b) <= 1e-10 * max(1.0, abs(b)), i


@pytest.mark.parametrize("path,k", [("skidpadoval", 0), ("vail", 1)])
def test_warm_nodes_agree(ctx, path, k):
    """Warm branch (coupled_lat_long.jl:82-102) from an exact first solve: both restatements interpolate the previous solution the same way,
    including the reference's prev_ts == ts aliasing (model_predictive_control.jl:15)."""
    traj, orc, T, st, ct, t0 = ctx[path]
    P, U = S.X1(), S.coupled_control_params()
    state, control, t = st[k], ct[k], float(t0[k])
    ts, dt = orc.time_steps(t)
    q1, u1, p1 = orc.nodes(state, control, ts, dt, time_offset=0.0)
    x, y, info = orc.solve_exact(orc.update_qp(q1, u1, p1, dt, state, control))
    X = orc.split_x(x)
    # second step 10 ms later from the same measured state
    ts2, dt2 = orc.time_steps(t + 0.01)
    q_o, u_o, p_o = orc.nodes(state, control, ts2, dt2, time_offset=0.0, solved=True, prev_ts=ts2, prev_q=X["q"], prev_u=X["u"])
    un = np.array([P["delta_max"], max(-P["Fx_min"], P["Fx_max"])])
    q_s, u_s, p_s = S.compute_linearization_nodes(P, U, T, state, control, ts2, dt2, NS, NL, time_offset=0.0, prev=(ts2, X["q"], X["u"], un))
    assert rel(q_s, q_o) < 1e-9 and rel(u_s, u_o) < 1e-9 and rel(p_s, p_o) < 1e-9


def test_pieces_agree_off_the_beaten_track(ctx):
    """Branches the synthetic batches rarely reach: saturated actuators (zero B columns), sliding tires, friction-limited seeding, s past the path end."""
    traj, orc, T, st, ct, t0 = ctx["skidpadoval"]
    P = S.X1()
    rng = np.random.default_rng(7)
    for _ in range(40):
        q = [rng.uniform(-1, 1), rng.uniform(2, 14), rng.uniform(-2, 2), rng.uniform(-1, 1), rng.uniform(-0.4, 0.4), rng.uniform(-1, 1)]
        u = [rng.uniform(-0.5, 0.5), rng.uniform(-22000, 9000)]            # beyond delta_max / Fx_min / Fx_max on purpose
        p = [rng.uniform(2, 14), rng.uniform(-0.1, 0.1), 0, 0]
        assert rel(S.tracking_vehicle_model(P, q, u, p), orc.tracking_dynamics(q, u, p)) < 1e-12
        A, B0, Bf, c = S.linearize(P, q, u, p[:2], u, p[:2], 0.2, False)
        Ao, B0o, Bfo, co = orc.linearize_interval(q, u, p, u, p, 0.2, False)
        assert rel(A, Ao) < 1e-9 and rel(B0, B0o) < 1e-9 and rel(c, co) < 1e-9
        if abs(u[0]) > P["delta_max"]:
            assert np.all(B0[:, 0] == 0.0)                                  # saturated steering: ForwardDiff sees a constant (vehicle_dynamics.jl:296)
    for V, At, kap, iters in [(12.0, 3.0, 0.08, 4), (14.0, -9.0, 0.01, 4), (8.0, 0.5, 0.2, 4), (3.0, 12.0, -0.3, 2)]:      # friction-saturated: :331-338
        a = S.steady_state_estimates(P, V, At, kap, num_iters=iters); b = orc.steady_state(V, At, kap, num_iters=iters)
        for key in ["beta", "Ux", "Uy", "r", "A", "delta", "Fxf", "Fxr"]:
            assert abs(a[key] - b[key]) <= 1e-9 * max(1.0, abs(b[key])), key
    s_end = float(traj.s[-1])
    for s in [s_end + 0.5, s_end + 25.0, -3.0]:                            # trajectories.jl:59 `s > traj.s[end]` and Line() extrapolation
        a = T.at_s(s); b = orc.traj_at_s(s)
        assert abs(a["V"] - b[2]) < 1e-10 and abs(a["psi"] - b[6]) < 1e-10 and abs(a["kappa"] - b[7]) < 1e-10 and abs(a["t"] - b[0]) < 1e-9


def test_inverse_tire_model_is_discontinuous_at_saturation():
    """A fact about the reference that every restatement inherits (found when the two restatements disagreed by O(1) on one friction-saturated seed):
    _invfialatiremodel (vehicle_dynamics.jl:56-62) returns -(3 Fy_max / C) sign(Fy) when |Fy| >= Fy_max but -(1 + cbrt(|Fy|/Fy_max - 1)) sign(Fy) --
    the slide RATIO, not tan(alpha) -- below it, so it jumps from ~1 to 3 Fy_max / C (~0.15 for X1) at |Fy| = Fy_max.  steady_state_estimates clamps
    the front force onto the friction circle and then evaluates it at exactly that point (:368-373: |Fyf| and Fyf_max are the same number up to rounding),
    so on a friction-saturated node the seeded delta is decided by the last bit.  Parity tests therefore exclude instances whose seed sits on the jump."""
    P = S.X1()
    Fy_max = 5000.0
    below = S._inv_fiala(Fy_max * (1 - 1e-15), P["Caf"], Fy_max); at = S._inv_fiala(Fy_max, P["Caf"], Fy_max)
    assert abs(below + 1.0) < 1e-4 and abs(at + 3 * Fy_max / P["Caf"]) < 1e-15 and abs(below - at) > 0.8
