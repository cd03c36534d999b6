"""GPU tests at BASELINE.json's full size (B = 4096): size-independent properties + oracle comparison on a random subset."""
import os

import numpy as np
import pytest

from conftest import ROOT, make_oracle

pytestmark = pytest.mark.gpu

B = 4096


@pytest.fixture(scope="module")
def run(pkg, skidpad):
    mpc = pkg.BatchedTrajectoryTrackingMPC(skidpad, B)
    state, control, t0, toff = pkg.synthetic.config2_inputs(skidpad, B, seed=12345)
    u, status, iters = mpc.step_(state, control, t0, time_offset=toff)
    return mpc, state, control, t0, toff, u, status, iters


def test_all_instances_solve(run, pkg):
    mpc, state, control, t0, toff, u, status, iters = run
    assert np.all(status == pkg.SOLVED)
    assert iters.max() <= 25 and np.all(np.isfinite(u))


def test_solution_satisfies_the_qp(run):
    """Properties that need no oracle: dynamics rows hold to rounding, every bound holds, complementary slack structure is sane."""
    mpc, state, control, t0, toff, u, status, iters = run
    qp = mpc.qp_data(); x, sg = mpc.solution()
    N = mpc.N
    o = 0
    A = qp[:, o:o + 36 * N].reshape(B, N, 6, 6); o += 36 * N
    B0 = qp[:, o:o + 12 * N].reshape(B, N, 6, 2); o += 12 * N
    Bf = qp[:, o:o + 12 * N].reshape(B, N, 6, 2); o += 12 * N
    c = qp[:, o:o + 6 * N].reshape(B, N, 6); o += 6 * N
    H = qp[:, o:o + 8 * N].reshape(B, N, 4, 2); o += 8 * N
    G = qp[:, o:o + 4 * N].reshape(B, N, 4); o += 4 * N
    dmin = qp[:, o:o + N]; o += N; dmax = qp[:, o:o + N]; o += N; fxmax = qp[:, o:o + N]; o += N; ddmin = qp[:, o:o + N]; o += N; ddmax = qp[:, o:o + N]; o += N
    q = x[:, :, :6]; un = x[:, :, 6:]
    pred = np.einsum("bkij,bkj->bki", A, q[:, :-1]) + np.einsum("bkij,bkj->bki", B0, un[:, :-1]) + np.einsum("bkij,bkj->bki", Bf, un[:, 1:]) + c
    assert np.max(np.abs(pred - q[:, 1:])) < 1e-9                      # C10 / C12
    tol = 1e-9
    assert np.all(un[:, 1:, 0] <= dmax + tol) and np.all(un[:, 1:, 0] >= dmin - tol) and np.all(un[:, 1:, 1] <= fxmax + tol)      # C13
    dd = np.diff(un[:, :, 0], axis=1)
    assert np.all(dd <= ddmax + tol) and np.all(dd >= ddmin - tol)
    assert np.all(q[:, :, 1] >= 1.0 - tol) and np.all(q[:, :, 1] <= 15.0 + tol) and np.all(un[:, :, 1] >= -1.0 - tol)              # C5-C7
    env = np.einsum("bkij,bkj->bki", H, q[:, 1:, 2:4]) - G
    assert np.all(env[:, :, :2] <= sg[:, :, 0:1] + tol) and np.all(env[:, :, 2:] <= sg[:, :, 1:2] + tol) and np.all(sg[:, :, :2] >= -tol)
    # hinge slacks are tight: sigma = max(0, max of its two rows)
    assert np.max(np.abs(sg[:, :, 0] - np.maximum(0, env[:, :, :2].max(2)))) < 1e-6
    assert np.max(np.abs(sg[:, :, 1] - np.maximum(0, env[:, :, 2:].max(2)))) < 1e-6
    # first node is the measured state (C8/C9)
    assert np.array_equal(x[:, 0, :6], qp[:, -11:-5]) and np.array_equal(x[:, 0, 6:], qp[:, -5:-3])


def test_random_subset_against_oracle(run, oracle_mod, skidpad):
    mpc, state, control, t0, toff, u, status, iters = run
    orc = make_oracle(oracle_mod, skidpad)
    rng = np.random.default_rng(99)
    idx = rng.choice(B, 48, replace=False)
    qp = mpc.qp_data(); x, _ = mpc.solution()
    worst = 0.0
    for b in idx:
        ts, dt = orc.time_steps(t0[b])
        qs, us, ps = orc.nodes(state[b], control[b], ts, dt, time_offset=toff[b])
        sd = orc.update_qp(qs, us, ps, dt, state[b], control[b])
        assert np.max(np.abs(sd - qp[b])) <= 1e-8 * max(1.0, np.max(np.abs(sd)))
        xe, ye, info = orc.solve_exact(qp[b])
        worst = max(worst, float(np.max(np.abs(x[b, 1, 6:] - orc.split_x(xe)["u"][1]))))
    assert worst < 1e-6, worst


def test_accuracy_distribution_over_the_batch(run, oracle_mod, skidpad):
    """Applied control against the exact optimum of the same QP data for every 4th instance of the 4096 batch (1024 oracle solves, threaded).
    Measured over all 4096 (tools/gpu_accuracy_full.py): median 2e-12, 99.9th percentile 2.5e-7, 3 instances between 1e-6 and 3e-6 -- nearly
    degenerate instances (a row whose slack and multiplier are both ~sqrt(mu)) approach the optimum like sqrt(mu) and the rounding noise of the
    Newton systems forbids mu < 1e-13.  The bar asserted here: 99.5 % within 1e-6, every instance within 1e-5."""
    from concurrent.futures import ThreadPoolExecutor
    mpc, state, control, t0, toff, u, status, iters = run
    qp = mpc.qp_data(); x, _ = mpc.solution()
    nthr = 8
    orcs = [make_oracle(oracle_mod, skidpad) for _ in range(nthr)]
    sel = np.arange(0, B, 4)

    def work(w):
        out = []
        for b in sel[w::nthr]:
            xe, ye, info = orcs[w].solve_exact(qp[b])
            out.append(float(np.max(np.abs(x[b, 1, 6:] - orcs[w].split_x(xe)["u"][1]))) if info["status"] == 1 else np.nan)
        return out
    with ThreadPoolExecutor(nthr) as ex:
        err = np.array(sum(ex.map(work, range(nthr)), []))
    assert not np.any(np.isnan(err))
    assert np.mean(err <= 1e-6) >= 0.995 and err.max() <= 1e-5 and np.median(err) <= 1e-9, (np.mean(err <= 1e-6), err.max(), np.median(err))


def test_golden_cases_on_gpu(pkg):
    """The committed vectors (tools/make_golden_cases.py): cold step and warm second step on two of the reference's test paths."""
    G = np.load(os.path.join(ROOT, "tests", "golden", "coupled_cases.npz"))
    for path in ["skidpadoval", "vail"]:
        traj = pkg.load_path_fixture(path)
        mpc = pkg.BatchedTrajectoryTrackingMPC(traj, 6)
        un = np.array([mpc.u_normalization[0], mpc.u_normalization[1], mpc.u_normalization[1]])
        u1, st, _ = mpc.step_(G[f"{path}_state"], G[f"{path}_control"], G[f"{path}_t0"], time_offset=G[f"{path}_toff"])
        assert np.all(st == 1)
        assert np.max(np.abs(mpc.qp_data() - G[f"{path}_qp1"]) / np.maximum(1.0, np.abs(G[f"{path}_qp1"]))) < 1e-8
        assert np.max(np.abs(u1 - G[f"{path}_u1"]) / un) < 1e-6
        u2, st, _ = mpc.step_(G[f"{path}_state2"], G[f"{path}_u1"], G[f"{path}_t0"] + 0.01, time_offset=G[f"{path}_toff"])
        assert np.all(st == 1)
        assert np.max(np.abs(u2 - G[f"{path}_u2"]) / un) < 1e-6


def test_ragged_and_tiny_batches(pkg, skidpad):
    """B = 1 and a batch that is not a multiple of the wave size give the same answers as the same instances inside a big batch."""
    state, control, t0, toff = pkg.synthetic.config2_inputs(skidpad, 70, seed=4)
    big = pkg.BatchedTrajectoryTrackingMPC(skidpad, 70)
    ub, st, _ = big.step_(state, control, t0, time_offset=toff)
    one = pkg.BatchedTrajectoryTrackingMPC(skidpad, 1)
    u1, st1, _ = one.step_(state[:1], control[:1], t0[:1], time_offset=toff[:1])
    assert st1[0] == 1 and np.array_equal(u1[0], ub[0])
    part = pkg.BatchedTrajectoryTrackingMPC(skidpad, 70)
    up, stp, _ = part.step_(state[:37], control[:37], t0[:37], time_offset=toff[:37])
    assert np.array_equal(up, ub[:37])
    with pytest.raises(pkg.PigeonError):
        big.step_(np.zeros((71, 6)), np.zeros((71, 3)), np.zeros(71))            # B > capacity is refused, not truncated


def test_closed_loop_on_device_matches_oracle(pkg, oracle_mod, skidpad):
    """pg_simulate_dev (simulate, model_predictive_control.jl:80-100: warm branch, one-step actuation delay, RK4 plant) against the oracle's
    closed loop with its exact solver, 40 steps x 24 instances; the two loops only share inputs."""
    Bc, steps = 24, 40
    state, control, t0, toff = pkg.synthetic.config2_inputs(skidpad, Bc, seed=77, traj_mode=False)
    mpc = pkg.BatchedTrajectoryTrackingMPC(skidpad, Bc)
    mpc.set_inputs(state, control, t0, time_offset=toff)
    s, c, t, qh, uh = mpc.simulate_(steps, dt=0.01, record=True)
    orc = make_oracle(oracle_mod, skidpad)
    q = state.copy(); u = control.copy(); tt = t0.copy()
    un = np.array([orc.u_norm[0], orc.u_norm[1], orc.u_norm[1]])
    for k in range(steps):
        assert np.max(np.abs(qh[k] - q) / np.maximum(1.0, np.abs(q))) < 1e-5, k
        assert np.max(np.abs(uh[k] - u) / un) < 1e-5, k
        unext, _, it, st, _ = orc.step_batch(q, u, tt, time_offsets=toff, solver=0)
        assert np.all(st == 1)
        q = np.stack([orc.plant_step(q[b], u[b], 0.01) for b in range(Bc)])
        u = unext; tt = tt + 0.01
    assert np.max(np.abs(s - q) / np.maximum(1.0, np.abs(q))) < 1e-5 and np.max(np.abs(c - u) / un) < 1e-5 and np.allclose(t, tt)
    st, it, act, mu = mpc.solve_info()
    assert np.all(st == 1)
