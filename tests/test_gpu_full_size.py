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


def test_accuracy_of_every_instance_of_the_batch(run, oracle_mod, skidpad, pkg):
    """BASELINE north star: controls within 1e-6 (rel-inf, normalised controls) of the reference on identical inputs, active-set indices bit-exact.
    Here for EVERY one of the 4096 config-2 instances (threaded oracle, exact optimum of the same QP data):
      * applied control and every control of the horizon <= 1e-6 (measured: 5e-11 / 4e-9, through the active-set rounds as through the interior point -- the active-set polish removed the sqrt(mu) tail that left
        3 of 4096 instances at 1e-6..3e-6 in round 1);
      * whole primal solution (states) <= 1e-6 relative;
      * the signed canonical active-set list IDENTICAL to the oracle's for every instance, under one rule applied on both sides: a row is active when its multiplier
        exceeds 1e-6 (a row at its bound with a zero multiplier is degenerate -- the QP does not define which side of "active" it falls on; pg_get_multipliers);
      * every instance polished (verified KKT point), none fell back to the interior-point iterate."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as om
    mpc, state, control, t0, toff, u, status, iters = run
    qp = mpc.qp_data(); x, _ = mpc.solution(); _, _, act, _ = mpc.solve_info(); pol = mpc.polish_info(); lam = mpc.multipliers()
    assert np.all(pol >= 1), np.bincount(pol + 1)
    nthr = min(16, len(os.sched_getaffinity(0)))
    orcs = [make_oracle(oracle_mod, skidpad) for _ in range(nthr)]

    def work(w):
        o = orcs[w]; out = []
        for b in range(w, B, nthr):
            xe, ye, info = o.solve_exact(qp[b]); X = o.split_x(xe)
            assert info["status"] == 1
            e_u2 = float(np.max(np.abs(x[b, 1, 6:] - X["u"][1]))); e_u = float(np.max(np.abs(x[b, :, 6:] - X["u"])))
            e_q = float(np.max(np.abs(x[b, :, :6] - X["q"]) / np.maximum(1.0, np.abs(X["q"]))))
            Q = o.assemble_qp(qp[b])
            # ONE rule on both sides: a row is active when its multiplier exceeds 1e-6 (mine: bit of the verified working set AND multiplier; theirs: multiplier)
            mine = set(mpc.canonical_active_set(b, act[b], qp[b], lam=lam[b])); theirs = set(om.active_set(Q, xe, ye, tol=1e-6))
            diff = mine ^ theirs
            # (a multiplier within 1e-9 of the threshold itself may fall on either side of it: not counted, reported)
            near = [i for i in diff if abs(abs(ye[abs(i) - 1]) - 1e-6) <= 1e-9]
            out.append((e_u2, e_u, e_q, len(diff) - len(near), len(near)))
        return out
    with ThreadPoolExecutor(nthr) as ex:
        res = np.array(sum(ex.map(work, range(nthr)), []))
    assert res.shape == (B, 5)
    assert res[:, 0].max() <= 1e-6 and res[:, 1].max() <= 1e-6 and res[:, 2].max() <= 1e-6, res.max(axis=0)
    assert np.median(res[:, 0]) <= 1e-11
    print(f"max |u2-u2*| {res[:, 0].max():.2e}, max |u-u*| {res[:, 1].max():.2e}, max rel |q-q*| {res[:, 2].max():.2e}, identical active-set lists {int((res[:, 3] == 0).sum())}/{B}"
          f" (rows whose multiplier sits within 1e-9 of the 1e-6 threshold: {int(res[:, 4].sum())})")
    assert res[:, 3].sum() == 0, (int(res[:, 3].sum()), np.flatnonzero(res[:, 3])[:8])          # the signed index lists are IDENTICAL for every one of the 4096 instances


def test_multipliers_of_held_rows_are_the_oracle_duals(run, oracle_mod, skidpad):
    """pg_get_multipliers against the oracle's dual solution, row by row, for 160 instances that needed more than one round (they hold steering-rate rows).  Since round 5 a
    held rate row is eliminated exactly in the recursion and its multiplier is READ OFF the stationarity condition in the pinned input (k_solve: polish_check) -- nothing
    iterates on it --, so this is the direct check of that expression: rows 12 / 13; next to them the bound rows held through the augmented Lagrangian (3, 4, 5) and the
    pivots of the eliminated slacks (10, 11), whose multipliers are the linear cost coefficients of their slacks.  Measured: rate rows 2.6e-8 at worst (median 6e-13)."""
    from concurrent.futures import ThreadPoolExecutor
    mpc, state, control, t0, toff, u, status, iters = run
    qp = mpc.qp_data(); _, _, act, _ = mpc.solve_info(); lam = mpc.multipliers(); pol = mpc.polish_info()
    N, Ns, Nl = mpc.N, mpc.N_short, mpc.N_long
    r_C1 = 0; r_C13 = 2 * N + Ns + 2 * N + 3 * (N + 1) + 6 + 2 + 6 * Ns + Ns + 6 * Nl      # (the row blocks of the canonical QP: mpc.canonical_active_set)
    pick = np.flatnonzero(pol >= 2)[:160]
    assert len(pick) == 160
    nthr = min(16, len(os.sched_getaffinity(0)))
    orcs = [make_oracle(oracle_mod, skidpad) for _ in range(nthr)]

    def work(w):
        out = []
        for b in pick[w::nthr]:
            xe, ye, info = orcs[w].solve_exact(qp[b])
            assert info["status"] == 1
            for k in range(N):
                base = r_C13 + 9 * k
                for j, row in ((3, base), (4, base + 1), (5, base + 2), (12, base + 7), (13, base + 8), (10, r_C1 + 2 * k), (11, r_C1 + 2 * k + 1)):
                    if (int(act[b, k]) >> j) & 1 and lam[b, k, j] > 1e-6:
                        out.append((j, abs(lam[b, k, j] - abs(ye[row])) / max(1.0, abs(ye[row]))))
        return out
    with ThreadPoolExecutor(nthr) as ex:
        res = np.array(sum(ex.map(work, range(nthr)), []))
    rate = res[(res[:, 0] == 12) | (res[:, 0] == 13)]
    assert len(rate) >= 300                                            # the sample does hold rate rows
    print(f"held rows compared: {len(res)} ({len(rate)} rate rows); worst |lambda - lambda*| / max(1, lambda*): rate rows {rate[:, 1].max():.1e}, all {res[:, 1].max():.1e}")
    assert res[:, 1].max() <= 1e-6, res[np.argmax(res[:, 1])]


def test_golden_cases_on_gpu(pkg):
    """The committed vectors (tools/make_golden_cases.py): cold step and warm second step on two of the reference's test paths."""
    G = np.load(os.path.join(ROOT, "tests", "golden", "coupled_cases.npz"))
    for path in ["skidpadoval", "vail"]:
        traj = pkg.load_path_fixture(path)
        mpc = pkg.BatchedTrajectoryTrackingMPC(traj, 6)
        un = np.array([mpc.u_normalization[0], mpc.u_normalization[1], mpc.u_normalization[1]])
        u1, st, _ = mpc.step_(G[f"{path}_state"], G[f"{path}_control"], G[f"{path}_t0"], time_offset=G[f"{path}_toff"])
        assert np.all(st == pkg.SOLVED)
        assert np.max(np.abs(mpc.qp_data() - G[f"{path}_qp1"]) / np.maximum(1.0, np.abs(G[f"{path}_qp1"]))) < 1e-8
        assert np.max(np.abs(u1 - G[f"{path}_u1"]) / un) < 1e-6
        u2, st, _ = mpc.step_(G[f"{path}_state2"], G[f"{path}_u1"], G[f"{path}_t0"] + 0.01, time_offset=G[f"{path}_toff"])
        assert np.all(st == pkg.SOLVED)
        assert np.max(np.abs(u2 - G[f"{path}_u2"]) / un) < 1e-6


def test_ragged_and_tiny_batches(pkg, skidpad):
    """B = 1 and a batch that is not a multiple of the wave size give the same answers as the same instances inside a big batch."""
    state, control, t0, toff = pkg.synthetic.config2_inputs(skidpad, 70, seed=4)
    big = pkg.BatchedTrajectoryTrackingMPC(skidpad, 70)
    ub, st, _ = big.step_(state, control, t0, time_offset=toff)
    one = pkg.BatchedTrajectoryTrackingMPC(skidpad, 1)
    u1, st1, _ = one.step_(state[:1], control[:1], t0[:1], time_offset=toff[:1])
    assert st1[0] == pkg.SOLVED and np.array_equal(u1[0], ub[0])
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
    # the loop's clock: `for t in 0:dt:trajectory.t[end]` (:87) shifted by each instance's start time, as Julia's range arithmetic gives it (oracle/julia_range.hpp)
    clock = np.stack([orc.simulate_times(0.01, float(skidpad.t[-1]), steps + 1, t_start=float(t0[b])) for b in range(Bc)], axis=1)
    for k in range(steps):
        assert np.max(np.abs(qh[k] - q) / np.maximum(1.0, np.abs(q))) < 1e-5, k
        assert np.max(np.abs(uh[k] - u) / un) < 1e-5, k
        unext, _, it, st, _ = orc.step_batch(q, u, tt, time_offsets=toff, solver=0)
        assert np.all(st == pkg.SOLVED)
        q = np.stack([orc.plant_step(q[b], u[b], 0.01) for b in range(Bc)])
        u = unext; tt = clock[k + 1]
    assert np.max(np.abs(s - q) / np.maximum(1.0, np.abs(q))) < 1e-5 and np.max(np.abs(c - u) / un) < 1e-5 and np.array_equal(t, tt)      # (times: bit for bit)
    st, it, act, mu = mpc.solve_info()
    assert np.all(st == pkg.SOLVED)


def test_warm_start_of_the_active_set_is_exact_and_mostly_sufficient(pkg, skidpad):
    """Closed loop on the device (pg_simulate_dev), 1024 instances: with the warm start of the active set (pg_config.warm_polish, the counterpart of the
    reference's OSQP warm start) an instance first tries the polish from the previous step's active set and multipliers.  A verified round is the exact optimum of
    the NEW QP, so after the first warm step the controls with and without it coincide for EVERY instance (both are exact solvers of the same QP); over twelve
    steps they stay together except where the closed loop itself is discontinuous (the reference's path projection flips segments at a vertex: an instance
    sitting on one amplifies 1e-12 into 1e-4); and nearly every instance is served by the warm polish alone (iters == 0)."""
    Bc = 1024
    state, control, t0, toff = pkg.synthetic.config2_inputs(skidpad, Bc, seed=31)
    un = np.array([0.314159, 16793.7, 16793.7])
    out = {}
    # three solvers of the same closed loop: warm start of the active set (the default); no warm start, every step from the empty set (pg_config.cold_guess);
    # no warm start and no guess, the interior point every step
    for key, warm, cg in (("warm", True, 8), ("guess", False, 8), ("ipm", False, 0)):
        mpc = pkg.BatchedTrajectoryTrackingMPC(skidpad, Bc, warm_polish=warm, cold_guess=cg)
        mpc.set_inputs(state, control, t0, time_offset=toff)
        mpc.simulate_(2)                                  # cold step, then ONE warm step
        u2 = mpc.get_next_control(); it2 = mpc.solve_info()[1]
        s, c, t, _, _ = mpc.simulate_(10)
        st, it, _, _ = mpc.solve_info(); pol = mpc.polish_info()
        assert np.all(st == pkg.SOLVED)
        out[key] = (s, c, it, pol, u2, it2)
        mpc.close()
    # first warm step, "warm" vs "guess": identical cold step, hence the same QP; same optimum for every instance
    assert np.max(np.abs(out["warm"][4] - out["guess"][4]) / un) < 1e-9
    # "warm" vs "ipm": their cold steps end in two verified KKT points ~1e-9 apart (different routes), the second step linearises about them
    d2 = np.max(np.abs(out["warm"][4] - out["ipm"][4]) / un, axis=1)
    assert np.median(d2) < 1e-9 and np.mean(d2 < 1e-6) > 0.99 and d2.max() < 1e-2, (np.median(d2), np.mean(d2 < 1e-6), d2.max())      # (max: same amplification as below)
    assert np.mean(out["warm"][5] == 0) > 0.8
    for other in ("guess", "ipm"):
        d = np.max(np.abs(out["warm"][1] - out[other][1]) / un, axis=1)
        # (two verified KKT points of one QP may differ by polish_tol / curvature ~ 1e-7 in the weakly determined far-horizon controls; the next step linearises about them)
        assert np.median(d) < 1e-9 and np.mean(d < 1e-6) > 0.95 and d.max() < 1e-2, (other, np.median(d), np.mean(d < 1e-6), d.max())
    assert np.mean(out["warm"][2] == 0) > 0.9 and np.all(out["warm"][3] >= 1)               # served by the warm polish, every instance a verified KKT point
    assert np.all(out["ipm"][2] > 0) and np.mean(out["guess"][2] == 0) > 0.8

def test_cold_guess_serves_most_instances_and_changes_no_answer(pkg, skidpad):
    """pg_config.cold_guess (default 8): a cold instance first tries the active-set polish from the EMPTY set.  A verified round is a KKT point of the QP, i.e. THE
    optimum (strictly convex in the controls), so the answers with and without the guess coincide; on config 2 no instance sees the interior
    point any more (iters == 0: > 99 % asserted), and every instance -- served by the guess or not -- ends as a verified KKT point."""
    B = 4096
    state, control, t0, toff = pkg.synthetic.config2_inputs(skidpad, B, seed=12345)
    un = np.array([0.314159, 16793.7, 16793.7])
    res = {}
    for cg in (0, 8):
        mpc = pkg.BatchedTrajectoryTrackingMPC(skidpad, B, cold_guess=cg)
        u, st, it = mpc.step_(state, control, t0, time_offset=toff)
        x, sg = mpc.solution(); pol = mpc.polish_info(); act = mpc.solve_info()[2]
        assert np.all(st == pkg.SOLVED) and np.all(pol >= 1)
        res[cg] = (u, it, x, act)
        mpc.close()
    assert np.all(res[0][1] > 0) and np.mean(res[8][1] == 0) > 0.99
    assert np.max(np.abs(res[8][0] - res[0][0]) / un) < 1e-8
    assert np.max(np.abs(res[8][2] - res[0][2])) < 1e-6                      # the whole primal solution, every node
    same = np.all(res[8][3] == res[0][3], axis=1)
    assert np.mean(same) > 0.97                                               # active sets identical except where a row is degenerate (active with a zero multiplier)


def test_fused_step_is_bit_identical(pkg, skidpad):
    """pg_set_fusion(1): update_QP! runs inside the solve kernel (the wave that solves an instance linearises it first).  Same device functions, so the QP data, the
    solution and the controls are bit-identical to the two-kernel sequence -- cold step and a warm step after it."""
    B = 2048
    state, control, t0, toff = pkg.synthetic.config2_inputs(skidpad, B, seed=77)
    out = {}
    for fused in (False, True):
        mpc = pkg.BatchedTrajectoryTrackingMPC(skidpad, B)
        mpc.set_fusion(fused)
        mpc.set_inputs(state, control, t0, time_offset=toff)
        mpc.step_dev(); mpc.synchronize()
        u1 = mpc.get_next_control().copy(); qp1 = mpc.qp_data().copy(); x1 = mpc.solution()[0].copy(); it1 = mpc.solve_info()[1].copy()
        mpc.step_dev(); mpc.synchronize()                     # warm step on the same inputs
        u2 = mpc.get_next_control().copy(); qp2 = mpc.qp_data().copy()
        out[fused] = (u1, qp1, x1, it1, u2, qp2)
        mpc.close()
    for a, b in zip(out[False], out[True]):
        assert np.array_equal(a, b)


def test_pipelined_nodes_and_update_qp_are_bit_identical(pkg, skidpad):
    """pg_set_pipeline (default on, 2304 <= B <= 8192 with cold instances): compute_linearization_nodes! and update_QP! run as one launch in which interval t is
    linearised as soon as nodes t, t + 1 are seeded.  Same device functions on the same arguments: nodes, QP data, solution and controls are bit-identical to the
    launch-per-phase sequence -- on a cold batch, on a MIXED batch (half the instances reset after a step: their wavefronts publish once, at the end) and with a
    ragged last wavefront (B not a multiple of 64)."""
    for Bn in (4096, 2304 + 37):
        state, control, t0, toff = pkg.synthetic.config2_inputs(skidpad, Bn, seed=5)
        out = {}
        for piped in (False, True):
            mpc = pkg.BatchedTrajectoryTrackingMPC(skidpad, Bn)
            mpc.set_pipeline(piped)
            mpc.set_inputs(state, control, t0, time_offset=toff)
            mpc.step_dev(); mpc.synchronize()
            r = [np.concatenate([a.reshape(Bn, -1) for a in mpc.nodes()], axis=1), mpc.qp_data().copy(), mpc.get_next_control().copy(), mpc.solution()[0].copy(), mpc.solve_info()[0].copy()]
            mask = np.zeros(Bn, bool); mask[::2] = True; mask[:64] = True; mask[64:128] = False
            mpc.reset(mask)                                   # a batch of cold and warm instances: one all-cold, one all-warm and many mixed wavefronts
            mpc.step_dev(); mpc.synchronize()
            r += [np.concatenate([a.reshape(Bn, -1) for a in mpc.nodes()], axis=1), mpc.qp_data().copy(), mpc.get_next_control().copy(), mpc.solve_info()[0].copy()]
            out[piped] = r
            # the path the test means to cover is the one that ran: both steps of the pipelined handle (cold, mixed) took the pipelined launch, none of the other handle's
            assert mpc.get_option("stat_pipelined_launches") == (2 if piped else 0) and mpc.pipeline_fallbacks() == 0
            mpc.close()
        assert np.all(out[True][4] == pkg.SOLVED)
        for a, b in zip(out[False], out[True]):
            assert np.array_equal(a, b, equal_nan=True)


def test_pipelined_launch_with_a_trajectory_library(pkg):
    """The pipelined launch with a LIBRARY of tubes and a per-instance selection (the un-staged instantiation: the searched channels stay in memory): bit-identical to
    the launch-per-phase sequence, at the smallest batch the pipeline serves (option "pipe_min", 2304 by default)."""
    paths = ["skidpadoval", "vail", "EastPaddock", "variable_speed"]
    tubes = [pkg.load_path_fixture(p) for p in paths]
    Bn = 2304
    idx = ((np.arange(Bn) * 7 + 3) % len(tubes)).astype(np.int32)
    state = np.zeros((Bn, 6)); control = np.zeros((Bn, 3)); t0 = np.zeros(Bn); toff = np.zeros(Bn)
    for k, t in enumerate(tubes):
        sel = np.where(idx == k)[0]
        s_, c_, t_, o_ = pkg.synthetic.config2_inputs(t, len(sel), seed=200 + k, s_range=None if t.s[-1] > 90 else (2.0, 0.4 * t.s[-1]))
        state[sel], control[sel], t0[sel], toff[sel] = s_, c_, t_, o_
    out = {}
    for piped in (False, True):
        mpc = pkg.BatchedTrajectoryTrackingMPC(tubes, Bn)
        mpc.set_trajectory_index(idx)
        mpc.set_pipeline(piped)
        mpc.set_inputs(state, control, t0, time_offset=toff)
        mpc.step_dev(); mpc.synchronize()
        out[piped] = [np.concatenate([a.reshape(Bn, -1) for a in mpc.nodes()], axis=1), mpc.qp_data().copy(), mpc.get_next_control().copy(), mpc.solve_info()[0].copy()]
        assert mpc.get_option("stat_pipelined_launches") == (1 if piped else 0) and mpc.pipeline_fallbacks() == 0
        mpc.close()
    assert np.mean(out[True][3] == pkg.SOLVED) > 0.99
    for a, b in zip(out[False], out[True]):
        assert np.array_equal(a, b, equal_nan=True)


def test_split_solve_launch_gives_the_same_answers(pkg, skidpad, monkeypatch):
    """k_solve as two launches (the rounds-only instantiation, then the full kernel in list mode over what it left: the default without a safety row) against the single
    kernel (option "solve_split" = 0), on the benchmark batch and on `vail` (two of whose 4096 cold instances need the interior point: the list-mode launch does real work there,
    and the next launch of that handle hands the whole batch to the full kernel -- decided on the device from the previous launch's count): same status, same interior-point
    iteration counts, controls of two verified KKT points of the same QP.  Instances served by their active-set rounds get the SAME BITS both ways (the checkpoint restart of
    the rounds-only instantiation does not change a bit), so an instance's answer does not depend on which launch shape its batch-mates caused."""
    B = 4096
    un = np.array([0.314159, 16793.7, 16793.7])
    for path in ("skidpadoval", "vail"):
        traj = pkg.load_path_fixture(path)
        kw = dict(s_range=(2.0, 0.4 * float(traj.s[-1]))) if float(traj.s[-1]) <= 100 else {}
        state, control, t0, toff = pkg.synthetic.config2_inputs(traj, B, seed=12345, **kw)
        out = {}
        for split in ("1", "0"):
            mpc = pkg.BatchedTrajectoryTrackingMPC(traj, B, options={"solve_split": int(split)})
            u1, st1, it1 = mpc.step_(state, control, t0, time_offset=toff)
            mpc.reset()
            u2, st2, it2 = mpc.step_(state, control, t0, time_offset=toff)          # (after a launch that left work: the split handle has switched to the single kernel)
            out[split] = (u1, st1, it1, u2, st2, it2, mpc.polish_info().copy())
            assert mpc.get_option("stat_split_solve_launches") == (2 if split == "1" else 0) and mpc.get_option("stat_single_solve_launches") == (0 if split == "1" else 2)
            mpc.close()
        a, b = out["1"], out["0"]
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and np.array_equal(a[4], b[4]) and np.array_equal(a[5], b[5]), path
        assert np.all(pkg.is_solved(a[1]))
        both = (a[6] >= 1) & (b[6] >= 1)
        assert np.max(np.abs(a[0][both] - b[0][both]) / un) < 1e-8 and np.max(np.abs(a[3][both] - b[3][both]) / un) < 1e-8, path
        rounds_only = (a[2] == 0) & (b[2] == 0)                  # served by active-set rounds in both launch shapes: bit for bit
        assert rounds_only.sum() >= B - 8 and np.array_equal(a[0][rounds_only], b[0][rounds_only]) and np.array_equal(a[3][rounds_only], b[3][rounds_only]), path
        if path == "vail":
            assert (a[2] > 0).sum() >= 1          # the list-mode launch had something to do


@pytest.mark.gpu
def test_round4_guess_checkpoint_and_lane_arrangement_change_nothing(pkg, skidpad, monkeypatch):
    """Round 4's three changes to the headline path, each against its switch, on the benchmark batch (B = 4096, cold):
    (i) option "lin_lanes": one lane per (instance, interval) with all eight tangent directions against the lane pair -- the QP data bit for bit (fp64);
    (ii) option "ck_riccati": the matrix recursion of a round restarted at its checkpoint -- controls, primal solution and statuses bit for bit;
    (iii) option "clip_guess": the rounds of a cold instance started from the clipped roll-out's working set against the plain empty set -- two routes to verified KKT
    points of the same QP: every instance solved without an interior-point iteration both ways, controls within 1e-8, the canonical active sets identical."""
    B = 4096
    un = np.array([0.314159, 16793.7, 16793.7])
    state, control, t0, toff = pkg.synthetic.config2_inputs(skidpad, B, seed=12345, traj_mode=True)

    def run(**opts):
        m = pkg.BatchedTrajectoryTrackingMPC(skidpad, B, options=opts)
        u, st, it = m.step_(state, control, t0, time_offset=toff)
        _, _, act, _ = m.solve_info()
        r = dict(u=u, st=st, it=it, x=m.solution()[0], qp=m.qp_data(), act=act, lam=m.multipliers(), pol=m.polish_info().copy())
        m.close()
        return r

    ref = run()
    assert np.all(ref["st"] == pkg.SOLVED) and np.all(ref["it"] == 0) and np.all(ref["pol"] >= 1)
    pair = run(lin_lanes=2)
    assert np.array_equal(pair["qp"], ref["qp"]) and np.array_equal(pair["u"], ref["u"])
    nock = run(ck_riccati=0)
    assert np.array_equal(nock["u"], ref["u"]) and np.array_equal(nock["x"], ref["x"]) and np.array_equal(nock["st"], ref["st"]) and np.array_equal(nock["pol"], ref["pol"])
    plain = run(clip_guess=0)
    assert np.all(plain["st"] == pkg.SOLVED) and np.all(plain["it"] == 0)
    assert np.max(np.abs(plain["u"] - ref["u"]) / un) < 1e-8
    can = lambda r: ((r["act"][:, :, None].astype(np.uint32) >> np.arange(16)[None, None, :]) & 1).astype(bool) & (r["lam"] > 1e-6)
    assert np.array_equal(can(plain), can(ref))
    assert ref["pol"].mean() < plain["pol"].mean() - 0.2            # (1.62 against 1.97 rounds per instance)
