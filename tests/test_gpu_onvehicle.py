"""GPU parity on the reference's OWN configuration and on the construction knobs no other test turns (tests/onvehicle_cases.py), against the committed oracle vectors
(tests/golden/onvehicle_cases.npz: tools/make_onvehicle_golden.py) -- nothing here runs the oracle.

  * Pigeon.jl:34-58: CoupledTrajectoryTrackingMPC(X1(), straight_trajectory(30., 5.), N_short=5, N_long=10) (with and without a safety-row grid installed) and
    DecoupledTrajectoryTrackingMPC(X1(), straight_trajectory(30., 5.)) from state (0, 0, 0, 5, 0, 0): the on-vehicle horizon on a TWO-node tube;
  * use_correction_step = false (model_predictive_control.jl:22-24), R_delta, R_Fx > 0 (coupled_lat_long.jl:36-37), N_HJI = 10, rk4_substeps = 4, dt_long = 0.1.

Bars: time grid bit-exact; nodes 1e-9; refreshed QP data 1e-8; applied control 1e-6 (normalised) of the exact optimum; active-set index lists identical under the canonical rule (multiplier > 1e-6 on both sides)."""
import os
import numpy as np
import pytest

import onvehicle_cases as oc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "onvehicle_cases.npz"))


def rel(a, g):
    a = np.asarray(a, float); g = np.asarray(g, float)
    return float(np.max(np.abs(a - g) / np.maximum(1.0, np.abs(g))))


@pytest.mark.parametrize("name", list(oc.CASES))
def test_case_matches_the_committed_oracle_vectors(pkg, golden, name):
    form, tname, kw, cp, hji = oc.CASES[name]
    G = {k: golden[f"{name}__{k}"] for k in ("state", "control", "t0", "toff", "other", "ts", "sep", "qs", "us", "ps", "sd", "u", "act")}
    traj = oc.trajectory(pkg, tname)
    if form == "coupled":
        params = pkg.CoupledControlParams(); params.update(cp)
        mpc = pkg.BatchedTrajectoryTrackingMPC(traj, oc.B, control_params=params, **kw)
        if hji:
            knots, V, g = pkg.synthetic.hji_grid(dims=oc.HJI_DIMS); mpc.set_hji_cache(knots, V, g)
    else:
        mpc = pkg.DecoupledTrajectoryTrackingMPC(pkg.X1(), traj, oc.B, **kw)
    u, status, iters = mpc.step_(G["state"], G["control"], G["t0"], other_car_state=G["other"] if hji else None, time_offset=G["toff"])
    assert np.all(pkg.is_solved(status)), status
    ts, dt, _ = mpc.time_steps()
    assert np.array_equal(ts, G["ts"])                                                    # T1 bit-exact, correction step on or off, dt_long 0.2 or 0.1
    qs, us, ps = mpc.nodes(); qp = mpc.qp_data(); st, it, act, mu = mpc.solve_info()
    if form == "coupled":
        sep = mpc.path_coordinates()
        assert np.max(np.abs(sep[:, :3] - G["sep"]) / np.maximum(1.0, np.abs(G["sep"]))) < 1e-9
        assert rel(qs, G["qs"]) < 1e-9 and rel(us, G["us"]) < 1e-9 and rel(ps[:, :, :2], G["ps"][:, :, :2]) < 1e-9
        assert rel(qp, G["sd"]) < 1e-8
        un = np.array([mpc.u_normalization[0], mpc.u_normalization[1], mpc.u_normalization[1]])
        assert np.max(np.abs(u - G["u"]) / un) < 1e-6, np.max(np.abs(u - G["u"]) / un)
        lam = mpc.multipliers()
        for b in range(oc.B):
            mine = set(mpc.canonical_active_set(b, act[b], qp[b], lam=lam[b])); theirs = set(G["act"][b, 1:1 + G["act"][b, 0]].tolist())
            assert mine == theirs, (b, sorted(mine ^ theirs, key=abs))                     # canonical rule on both sides: multiplier > 1e-6 (tools/make_onvehicle_golden.py: tol = 1e-6)
    else:
        from oracle import oracle as om      # (only its layout helpers: nothing is solved here)
        o = om.OracleDecoupled(**kw)
        assert rel(qs[:, :, 2:6], G["qs"]) < 1e-9 and np.max(np.abs(us - G["us"]) / np.maximum(1.0, np.abs(G["us"]))) < 1e-9
        for b in range(oc.B):
            ref = om.embed_sd(o, G["sd"][b])
            assert rel(qp[b], ref) < 1e-8, b
        assert np.max(np.abs(u[:, 0] - G["u"][:, 0])) < 1e-6 and rel(u[:, 1:], G["u"][:, 1:]) < 1e-9
    if name.startswith("singleton"):
        assert abs(u[0, 0]) < 1e-9                                                         # the dry run itself: straight ahead
    mpc.close()
