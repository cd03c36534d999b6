"""GPU parity sweep over every path the reference ships (test/path/*.world, committed as data under tests/golden/paths) in both tracking
modes (trajectory mode: time_offset = 0; path mode: time_offset = NaN, ros_integration.jl:74-76), cold and warm.  Per path and mode: every
instance solves; a subsample is compared with the oracle stage by stage (nodes 1e-9, QP data 1e-8, applied control 1e-6 against the exact
optimum of the same QP data, identical active-set index lists)."""
import os

import numpy as np
import pytest

from conftest import ROOT, make_oracle

pytestmark = pytest.mark.gpu

PATHS = sorted(f[:-4] for f in os.listdir(os.path.join(ROOT, "tests", "golden", "paths")) if f.endswith(".npz")) + ["raw:curvy.world", "raw:curvy.msg"]
B = 384


def load_tube(pkg, path):
    """npz fixtures (decoded channels) or, for the raw data files, the product's own ingest (TrajectoryTube.from_world / from_path_msg)."""
    if path.startswith("raw:"):
        f = os.path.join(ROOT, "tests", "golden", "raw", path[4:])
        return pkg.TrajectoryTube.from_world(f) if f.endswith(".world") else pkg.TrajectoryTube.from_path_msg(f)
    return pkg.load_path_fixture(path)


def rel_inf(a, b, floor=1.0):
    a = np.asarray(a); b = np.asarray(b)
    return float(np.max(np.abs(a - b)) / max(floor, float(np.max(np.abs(b)))))


@pytest.mark.parametrize("traj_mode", [True, False], ids=["traj", "path"])
@pytest.mark.parametrize("path", PATHS)
def test_path_sweep(pkg, oracle_mod, path, traj_mode):
    tube = load_tube(pkg, path)
    s_range = None if tube.s[-1] > 90 else (2.0, 0.4 * tube.s[-1])
    state, control, t0, toff = pkg.synthetic.config2_inputs(tube, B, seed=sum(map(ord, path)) % 1000, traj_mode=traj_mode, s_range=s_range)
    mpc = pkg.BatchedTrajectoryTrackingMPC(tube, B)
    orc = make_oracle(oracle_mod, tube)
    u, status, iters = mpc.step_(state, control, t0, time_offset=toff)
    # `curvy` (the raw-file path) asks for 10 m/s through curvature up to 1/m: V^2 kappa = 100 m/s^2, ten times what friction allows.  A large share of its
    # QPs is INFEASIBLE (steering bounds and rate limits against the linearised dynamics); the reference's OSQP would report that and the ROS loop ignores
    # the status (ros_integration.jl:127).  There the check is: the solved instances agree with the oracle as everywhere else, and an instance this library
    # gives up on is one the oracle's exact solver cannot solve either.
    stress = path.startswith("raw:")
    if stress:
        assert np.mean(status == pkg.SOLVED) > 0.3, (path, np.bincount(status))
    else:
        assert np.all(status == pkg.SOLVED), (path, np.bincount(status))
    qs, us, ps = mpc.nodes(); qp = mpc.qp_data(); x, _ = mpc.solution(); _, _, act, _ = mpc.solve_info(); lam = mpc.multipliers()
    n_unsolved_checked = 0
    for b in range(0, B, 48 if not stress else 16):
        if status[b] != 1:
            if n_unsolved_checked < 2:           # (the oracle's fallback for such QPs is a 400k-iteration ADMM: check a couple, not all)
                xe, ye, info = orc.solve_exact(qp[b])
                assert info["status"] != 1, (path, b, "given up on a QP the oracle solves")
                n_unsolved_checked += 1
            continue
        if stress:
            # friction-saturated seeds sit on the jump of the reference's inverse tire model (tests/test_spec_numpy.py): the seeded steering there is decided
            # by the last bit, so on this path update_QP! and the solve are checked from the library's own nodes and the seeds only where they are well defined
            ts, dt = orc.time_steps(t0[b])
            oq, ou, op = orc.nodes(state[b], control[b], ts, dt, time_offset=toff[b])
            assert rel_inf(qs[b][:11], oq[:11]) < 1e-9 and rel_inf(ps[b][:11], op[:11]) < 1e-9, (path, b)
            sd = orc.update_qp(qs[b], us[b], ps[b], dt, state[b], control[b], (0, 0, 0, 0))
            assert rel_inf(qp[b], sd) < 1e-8, (path, b)
            xe, ye, info = orc.solve_exact(qp[b])
            assert info["status"] == 1 and rel_inf(x[b, 1, 6:], orc.split_x(xe)["u"][1]) < 1e-6, (path, b)
            continue
        ts, dt = orc.time_steps(t0[b])
        oq, ou, op = orc.nodes(state[b], control[b], ts, dt, time_offset=toff[b])
        assert rel_inf(qs[b], oq) < 1e-9 and rel_inf(us[b], ou) < 1e-9 and rel_inf(ps[b], op) < 1e-9, (path, b)
        sd = orc.update_qp(oq, ou, op, dt, state[b], control[b], (0, 0, 0, 0))
        G = orc.unpack_sd(qp[b]); O = orc.unpack_sd(sd)
        for k in O:
            assert rel_inf(G[k], O[k]) < 1e-8, (path, b, k)
        xe, ye, info = orc.solve_exact(qp[b])
        assert info["status"] == 1
        assert rel_inf(x[b, 1, 6:], orc.split_x(xe)["u"][1]) < 1e-6, (path, b)
        assert mpc.canonical_active_set(b, act[b], qp[b], lam=lam[b]) == oracle_mod.active_set(orc.assemble_qp(qp[b]), xe, ye, tol=1e-6), (path, b)
    # second (warm) step on the same handle: plant advanced by the oracle's model with the old control (simulate semantics)
    sel = np.arange(0, B, 48)
    state2 = state.copy()
    for b in sel:
        state2[b] = orc.plant_step(state[b], control[b], 0.01)
    u2, status2, _ = mpc.step_(state2, u, t0 + 0.01, time_offset=toff)
    if not stress:
        assert np.all(status2[sel] == pkg.SOLVED)
    qp2 = mpc.qp_data(); x2, _ = mpc.solution()
    for b in [b for b in sel if status2[b] == pkg.SOLVED][:3]:
        xe, ye, info = orc.solve_exact(qp2[b])
        assert rel_inf(x2[b, 1, 6:], orc.split_x(xe)["u"][1]) < 1e-6, (path, b, "warm")
    mpc.close()
