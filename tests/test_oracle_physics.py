"""CPU tests of the oracle's physics restatement (self-consistency; the reference holds no vectors for any of this)."""
import math

import numpy as np
import pytest

from conftest import make_oracle


@pytest.fixture(scope="module")
def orc(oracle_mod, skidpad):
    return make_oracle(oracle_mod, skidpad)


def test_x1_constants(orc, pkg):
    """vehicles.jl:1-59 / SURVEY.md D7."""
    v = orc.vehicle()
    assert v["m"] == 1964.0 and v["L"] == 2.87 and abs(v["a"] - 1.49784) < 1e-5 and abs(v["b"] - 1.37216) < 1e-5
    assert abs(v["h"] - 0.47) < 1e-12 and abs(v["Fx_min"] + 16793.733) < 1e-3 and abs(v["delta_max"] - 0.314159) < 1e-6
    assert abs(v["kappa_max"] - 0.113212) < 1e-6
    assert np.allclose(orc.u_norm, [0.314159265, 16793.733], rtol=1e-6)
    X = pkg.X1()
    for k, val in v.items():
        assert X[k] == pytest.approx(val, rel=1e-15), k


def test_time_steps_properties(orc):
    """model_predictive_control.jl:17-30: first long step in [dt_short, dt_long + dt_short); long nodes on the dt_long lattice."""
    rng = np.random.default_rng(0)
    for t0 in rng.uniform(0, 50, 200):
        ts, dt = orc.time_steps(t0)
        assert np.allclose(dt[:10], 0.01, atol=1e-12) and np.allclose(dt[11:], 0.2, atol=1e-12)
        assert 0.01 - 1e-12 <= dt[10] < 0.21 + 1e-12
        k = ts[11:] / 0.2
        assert np.allclose(k, np.round(k), atol=1e-9)
        assert ts[0] == t0


def test_adiff_and_relative_state(orc):
    x = orc.hji_relative_state([1.0, 2.0, 0.3, 5.0, 0.1, 0.02], [4.0, 6.0, 0.3 + 2 * math.pi + 0.2, 3.0])
    assert abs(x[2] - 0.2) < 1e-12 and x[3] == 5.0 and x[5] == 3.0 and x[6] == 0.02
    # HJI_computation.jl:21-22 with psi measured from North: (dE,dN) -> ego frame
    c, s = math.sin(-0.3), math.cos(-0.3)
    assert abs(x[0] - (c * 3 + s * 4)) < 1e-12 and abs(x[1] - (-s * 3 + c * 4)) < 1e-12


def test_tire_and_dynamics_sanity(orc):
    # straight driving at constant speed: drag only
    v = orc.vehicle()
    d = orc.tracking_dynamics([0, 6.0, 0, 0, 0, 0], [0.0, 0.0], [6.0, 0, 0, 0])
    assert abs(d[0]) < 1e-12 and abs(d[1] - (-(v["Cd0"] + 6 * v["Cd1"]) / v["m"])) < 1e-12 and np.allclose(d[2:], 0, atol=1e-12)
    # left steer => positive lateral acceleration and yaw acceleration
    d = orc.tracking_dynamics([0, 6.0, 0, 0, 0, 0], [0.05, 0.0], [6.0, 0, 0, 0])
    assert d[2] > 0 and d[3] > 0
    # actuator limits (vehicle_dynamics.jl:293-298): saturated steering has zero sensitivity
    d1 = orc.tracking_dynamics([0, 6.0, 0, 0, 0, 0], [0.5, 0.0], [6.0, 0, 0, 0])
    d2 = orc.tracking_dynamics([0, 6.0, 0, 0, 0, 0], [0.4, 0.0], [6.0, 0, 0, 0])
    assert np.array_equal(d1, d2)


def test_linearize_matches_finite_differences(orc):
    """A, B0, Bf, c of the restated `linearize` against central differences of the restated `propagate` (same RK4)."""
    rng = np.random.default_rng(1)
    for ramp in (False, True):
        q = np.array([0.1, 6.0, 0.1, 0.05, 0.02, 0.2]) + rng.normal(0, 0.01, 6)
        u0 = np.array([0.03, 300.0]); uf = np.array([0.05, -200.0]); p0 = np.array([6.0, 0.02, 0, 0]); pf = np.array([6.1, 0.03, 0, 0])
        dt = 0.2 if ramp else 0.01
        A, B0, Bf, c = orc.linearize_interval(q, u0, p0, uf, pf, dt, ramp)
        phi = orc.propagate_tracking(q, u0, p0, uf, pf, dt, ramp)
        assert np.allclose(A @ q + B0 @ u0 + (Bf @ uf if ramp else 0) + c, phi, atol=1e-12)
        for j in range(6):
            h = 1e-6; e = np.zeros(6); e[j] = h
            fd = (orc.propagate_tracking(q + e, u0, p0, uf, pf, dt, ramp) - orc.propagate_tracking(q - e, u0, p0, uf, pf, dt, ramp)) / (2 * h)
            assert np.allclose(A[:, j], fd, atol=2e-8), (ramp, j)
        for j in range(2):
            h = 1e-6 * (1.0 if j == 0 else 1e3); e = np.zeros(2); e[j] = h
            fd = (orc.propagate_tracking(q, u0 + e, p0, uf, pf, dt, ramp) - orc.propagate_tracking(q, u0 - e, p0, uf, pf, dt, ramp)) / (2 * h)
            assert np.allclose(B0[:, j], fd, atol=2e-8)
            if ramp:
                fd = (orc.propagate_tracking(q, u0, p0, uf + e, pf, dt, ramp) - orc.propagate_tracking(q, u0, p0, uf - e, pf, dt, ramp)) / (2 * h)
                assert np.allclose(Bf[:, j], fd, atol=2e-8)
            else:
                assert np.all(Bf == 0)


def test_trajectory_lookups(orc, skidpad):
    """traj(t), traj[s], path_coordinates (trajectories.jl:47-94) on the constant-speed skidpad oval."""
    s = 37.3
    n = orc.traj_at_s(s)
    assert abs(n[1] - s) < 1e-12 and abs(n[2] - 6.0) < 1e-12 and abs(n[0] - s / 6.0) < 1e-9
    n2 = orc.traj_at_time(n[0])
    assert abs(n2[1] - s) < 1e-9
    E, N, psi = n[4], n[5], n[6]
    e = 0.25
    s_, e_, t_, i = orc.path_coordinates(E - e * math.cos(psi), N - e * math.sin(psi))
    assert abs(s_ - s) < 2e-3 and abs(e_ - e) < 1e-4 and abs(t_ - s_ / 6.0) < 1e-9
    # brute-force projection in numpy: same segment index (strict '<' => lowest index)
    P = np.stack([skidpad.E, skidpad.N], 1); x = np.array([E - e * math.cos(psi), N - e * math.sin(psi)])
    v = P[1:] - P[:-1]; lam = np.clip(np.sum(v * (x - P[:-1]), 1) / np.sum(v * v, 1), 0, 1)
    d2 = np.sum(((1 - lam)[:, None] * P[:-1] + lam[:, None] * P[1:] - x) ** 2, 1)
    assert int(np.argmin(d2)) == i


def test_hji_lookup_multilinear(orc, pkg):
    knots, V, g = pkg.synthetic.hji_grid(dims=(5, 5, 4, 4, 4, 4, 4))
    orc.set_hji_grid(knots, V, g)
    rng = np.random.default_rng(2)
    # exact at grid nodes
    idx = [2, 1, 3, 0, 2, 1, 3]
    x = [float(knots[d][idx[d]]) for d in range(7)]
    Vq, gq, inb = orc.hji_lookup(x)
    lin = sum(idx[d] * int(np.prod([len(knots[k]) for k in range(d)])) for d in range(7))
    assert inb and abs(Vq - V[lin]) < 1e-6 and np.allclose(gq, g[lin], atol=1e-6)
    # scipy cross-check in the interior
    from scipy.interpolate import RegularGridInterpolator
    shape = tuple(len(k) for k in knots)
    rgi = RegularGridInterpolator([k.astype(np.float64) for k in knots], V.reshape(shape, order="F").astype(np.float64))
    for _ in range(20):
        x = np.array([rng.uniform(k[0], k[-1]) for k in knots], dtype=np.float64)
        Vq, gq, inb = orc.hji_lookup(x)
        assert inb and abs(Vq - float(rgi(x)[0])) < 1e-9
    # out of bounds => (Inf, 0): HJI_computation.jl:70
    x[0] = knots[0][-1] + 1.0
    Vq, gq, inb = orc.hji_lookup(x)
    assert (not inb) and math.isinf(Vq) and np.all(gq == 0)


def test_hji_constraint_branches(orc, pkg):
    knots, V, g = pkg.synthetic.hji_grid(dims=(5, 5, 4, 4, 4, 4, 4))
    orc.set_hji_grid(knots, V, g)
    ego = np.array([0.0, 0.0, 0.0, 6.0, 0.0, 0.0])
    far = np.array([0.0, 100.0, 0.0, 5.0])        # outside the grid => inactive row (M = 0, b = 1)
    M, b, Vv = orc.hji_constraint(ego, far, [0.0, 0.0, 0.0])
    assert np.all(M == 0) and b == 1.0 and math.isinf(Vv)
    near = np.array([0.5, 2.0, 0.1, 5.0])         # V <= eps => active row with finite data
    orc.set_hji_eps(10.0)
    M, b, Vv = orc.hji_constraint(ego, near, [0.01, 0.0, 100.0])
    assert np.all(np.isfinite(M)) and np.isfinite(b) and Vv <= 10.0
    # H4 hazard of the reference: other-car speed 0 inside the grid => NaN
    M, b, Vv = orc.hji_constraint(ego, [0.5, 2.0, 0.1, 0.0], [0.01, 0.0, 100.0])
    assert math.isnan(b)
    orc.set_hji_eps(0.05)
