"""CPU: the committed vectors of the reference's own dry-run configuration and of the construction knobs (tests/golden/onvehicle_cases.npz, written by
tools/make_onvehicle_golden.py) are what the oracle produces today (drift pin), and the independent numpy specification agrees with them on the refreshed QP data."""
import os
import numpy as np
import pytest

import onvehicle_cases as oc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "onvehicle_cases.npz"))


def test_singleton_is_the_reference_dry_run(golden):
    """Pigeon.jl:34-58: state (0, 0, 0, 5, 0, 0), zero control, t = 0, path mode; n = 193, m = 351 for N_short = 5, N_long = 10."""
    for name in ("singleton_coupled", "singleton_decoupled"):
        assert np.array_equal(golden[f"{name}__state"][0], [0, 0, 0, 5, 0, 0]) and np.all(golden[f"{name}__control"][0] == 0) and golden[f"{name}__t0"][0] == 0
        assert np.isnan(golden[f"{name}__toff"][0])
    assert golden["singleton_coupled__ts"].shape == (oc.B, 16) and golden["singleton_decoupled__ts"].shape == (oc.B, 31)
    # on the straight tube at its own speed the dry run's optimum is "keep going straight": zero steering, a small drive force against the drag
    u = golden["singleton_coupled__u"][0]
    assert abs(u[0]) < 1e-12 and u[1] == 0 and 0 < u[2] < 100


@pytest.mark.parametrize("name", list(oc.CASES))
def test_oracle_reproduces_the_golden_vectors(pkg, oracle_mod, golden, name):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_onvehicle_golden as mk
    seed = 500 + list(oc.CASES).index(name)
    r = mk.run_case(pkg, name, seed)
    for k in ("state", "control", "t0", "toff", "other", "ts"):
        assert np.array_equal(r[k], golden[f"{name}__{k}"], equal_nan=True), k
    for k, tol in (("qs", 1e-12), ("us", 1e-12), ("ps", 1e-12), ("sd", 1e-11), ("u", 1e-7)):
        g = golden[f"{name}__{k}"]
        assert np.max(np.abs(r[k] - g) / np.maximum(1.0, np.abs(g))) <= tol, k
    assert np.array_equal(r["act"], golden[f"{name}__act"])


@pytest.mark.parametrize("name", [n for n, c in oc.CASES.items() if c[0] == "coupled" and not c[4]])
def test_numpy_specification_agrees_on_the_knob_cases(pkg, golden, name):
    """oracle/spec_numpy.py (written from the Julia files independently of the C++) on the same inputs: time grid bit-exact, nodes 1e-9, refreshed QP data 1e-9."""
    from oracle import spec_numpy as S
    from oracle import oracle as om
    form, tname, kw, cp, hji = oc.CASES[name]
    traj = oc.trajectory(pkg, tname); T = S.Trajectory(traj.data)
    P, U = S.X1(), S.coupled_control_params(); U.update(cp)
    Ns, Nl = kw["N_short"], kw["N_long"]
    o = om.Oracle(**kw)
    for b in range(oc.B):
        state, control, t0, toff = golden[f"{name}__state"][b], golden[f"{name}__control"][b], float(golden[f"{name}__t0"][b]), float(golden[f"{name}__toff"][b])
        ts, dt = S.compute_time_steps(t0, N_short=Ns, N_long=Nl, dt_long=kw.get("dt_long", 0.2), use_correction_step=kw.get("use_correction_step", True))
        assert np.array_equal(ts, golden[f"{name}__ts"][b])
        q, u, p = S.compute_linearization_nodes(P, U, T, state, control, ts, dt, Ns, Nl, time_offset=toff)
        gq, gu, gp = golden[f"{name}__qs"][b], golden[f"{name}__us"][b], golden[f"{name}__ps"][b]
        rel = lambda a, g: float(np.max(np.abs(np.asarray(a) - g) / np.maximum(1.0, np.abs(g))))
        assert rel(q, gq) < 1e-9 and rel(u, gu) < 1e-9 and rel(p, gp) < 1e-9, (b, rel(q, gq), rel(u, gu))
        D = S.update_qp(P, U, gq, gu, gp, dt, Ns, Nl, nsub=kw.get("rk4_substeps", 10))
        G = o.unpack_sd(golden[f"{name}__sd"][b])
        for key in ("A", "B0", "Bf", "c", "H", "G", "dmin", "dmax", "fxmax", "ddmin", "ddmax"):
            assert rel(np.asarray(D[key]).reshape(G[key].shape), G[key]) < 1e-9, (b, key)
