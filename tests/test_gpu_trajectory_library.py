"""GPU parity for (x0, reference-trajectory) instances: one handle carries a LIBRARY of TrajectoryTubes (the reference's test paths, ragged lengths)
and a per-instance selection; every instance must reproduce what a single-trajectory controller on ITS tube computes (oracle, one per tube).
Also covers the un-staged nodes kernel (a tube longer than the LDS staging limit of 2048 nodes)."""
import numpy as np
import pytest

from conftest import make_oracle

pytestmark = pytest.mark.gpu

PATHS = ["skidpadoval", "vail", "EastPaddock", "variable_speed"]
B = 160


def rel_inf(a, b, floor=1.0):
    a = np.asarray(a); b = np.asarray(b)
    return float(np.max(np.abs(a - b)) / max(floor, float(np.max(np.abs(b)))))


@pytest.fixture(scope="module")
def library(pkg, oracle_mod):
    tubes = [pkg.load_path_fixture(p) for p in PATHS]
    assert len({len(t) for t in tubes}) > 1                      # ragged on purpose
    idx = (np.arange(B) * 7 + 3) % len(tubes)
    state = np.zeros((B, 6)); control = np.zeros((B, 3)); t0 = np.zeros(B); toff = np.zeros(B)
    for k, t in enumerate(tubes):
        sel = np.where(idx == k)[0]
        s_, c_, t_, o_ = pkg.synthetic.config2_inputs(t, len(sel), seed=100 + k, s_range=None if t.s[-1] > 90 else (2.0, 0.4 * t.s[-1]))
        state[sel], control[sel], t0[sel], toff[sel] = s_, c_, t_, o_
    orcs = [make_oracle(oracle_mod, t) for t in tubes]
    return tubes, idx.astype(np.int32), state, control, t0, toff, orcs


def test_index_is_required(pkg, library):
    tubes, idx, state, control, t0, toff, _ = library
    mpc = pkg.BatchedTrajectoryTrackingMPC(tubes, B)
    mpc.set_inputs(state, control, t0, time_offset=toff)
    with pytest.raises(pkg.PigeonError):
        mpc.compute_time_steps_()
    with pytest.raises(pkg.PigeonError):
        mpc.set_trajectory_index(np.full(B, len(tubes), dtype=np.int32))     # out of range
    mpc.close()


def test_library_step_matches_single_trajectory_oracles(pkg, oracle_mod, library):
    tubes, idx, state, control, t0, toff, orcs = library
    mpc = pkg.BatchedTrajectoryTrackingMPC(tubes, B)
    mpc.set_trajectory_index(idx)
    u, status, _ = mpc.step_(state, control, t0, time_offset=toff)
    assert np.all(status == pkg.SOLVED), status
    sep = mpc.path_coordinates(); qs, us, ps = mpc.nodes(); qp = mpc.qp_data(); x, _ = mpc.solution()
    for b in range(B):
        orc = orcs[idx[b]]
        s, e, t, _ = orc.path_coordinates(state[b, 0], state[b, 1])
        assert abs(sep[b, 0] - s) <= 1e-9 * max(1, abs(s)) and abs(sep[b, 1] - e) <= 1e-9 and abs(sep[b, 2] - t) <= 1e-9 * max(1, abs(t)), b
        ts, dt = orc.time_steps(t0[b])
        oq, ou, op = orc.nodes(state[b], control[b], ts, dt, time_offset=toff[b])
        assert rel_inf(qs[b], oq) < 1e-9 and rel_inf(us[b], ou) < 1e-9 and rel_inf(ps[b], op) < 1e-9, b
        sd = orc.update_qp(oq, ou, op, dt, state[b], control[b], (0, 0, 0, 0))
        G = orc.unpack_sd(qp[b]); O = orc.unpack_sd(sd)
        for k in O:
            assert rel_inf(G[k], O[k]) < 1e-8, (b, k)
        if b % 4 == 0:
            xe, ye, info = orc.solve_exact(qp[b])
            assert info["status"] == 1
            assert rel_inf(x[b, 1, 6:], orc.split_x(xe)["u"][1]) < 1e-6, b
    # the same instances through four single-trajectory handles give bit-identical controls
    for k, t in enumerate(tubes):
        sel = np.where(idx == k)[0]
        one = pkg.BatchedTrajectoryTrackingMPC(t, len(sel))
        u1, st1, _ = one.step_(state[sel], control[sel], t0[sel], time_offset=toff[sel])
        assert np.array_equal(u1, u[sel]), k
        one.close()
    mpc.close()


def test_long_tube_unstaged_nodes_kernel(pkg, oracle_mod, library):
    """A tube with more than 2048 nodes takes the global-memory search path of k_nodes; results must not depend on it."""
    tubes = library[0]
    t = tubes[0]
    # densify: insert two interior nodes per segment by the tube's own constant-acceleration law so the refined tube describes the same motion
    s = t.s; fine_s = np.unique(np.concatenate([s, s[:-1] + np.diff(s) / 3, s[:-1] + 2 * np.diff(s) / 3]))
    def lerp(ch): return np.interp(fine_s, s, ch)
    dense = pkg.TrajectoryTube(lerp(t.t), fine_s, lerp(t.V), lerp(t.A), lerp(t.E), lerp(t.N), lerp(t.psi), lerp(t.kappa))
    assert len(dense) > 2048
    n = 64
    state, control, t0, toff = pkg.synthetic.config2_inputs(dense, n, seed=5)
    mpc = pkg.BatchedTrajectoryTrackingMPC(dense, n)
    u, status, _ = mpc.step_(state, control, t0, time_offset=toff)
    assert np.all(status == pkg.SOLVED)
    qs, us, ps = mpc.nodes()
    orc = make_oracle(oracle_mod, dense)
    for b in range(n):
        ts, dt = orc.time_steps(t0[b])
        oq, ou, op = orc.nodes(state[b], control[b], ts, dt, time_offset=toff[b])
        assert rel_inf(qs[b], oq) < 1e-9 and rel_inf(us[b], ou) < 1e-9 and rel_inf(ps[b], op) < 1e-9, b
    mpc.close()
