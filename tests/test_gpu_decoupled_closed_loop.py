"""The decoupled (lateral) formulation in CLOSED LOOP on the device and its warm start (k_solve_lat).

Reference: the lateral QP is built with OSQPSettings.WarmStart() = true (decoupled_lat_long.jl:139) and runs in the same loop as the coupled one
(Pigeon.jl:34 X1DMPC, ros_integration.jl:94-103); `simulate` is generic over both formulations (model_predictive_control.jl:80-100).  Here: pg_simulate_dev on a
PG_DECOUPLED handle against an oracle loop (OracleDecoupled nodes / update_QP / exact verified optimum / get_next_control + the oracle's RK4 plant) that shares nothing
with it but the inputs, and the warm step of the full benchmark batch, every instance against the exact optimum of its own QP data."""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from test_gpu_decoupled import check_lateral_batch_against_oracle

pytestmark = pytest.mark.gpu
UN = np.array([0.314159, 16793.7, 16793.7])        # (delta_max, Fx scale): the normalisation the coupled closed-loop test states its bar in


def oracle_lateral_loop(oracle_mod, tube, Ns, Nl, walls, Ww, state, control, t0, toff, steps, dt=0.01):
    """simulate (model_predictive_control.jl:80-100) with the lateral oracle, per instance: returns the histories pushed at :88-89 and the final (q, u)."""
    B = len(t0)
    nthr = min(16, len(os.sched_getaffinity(0)), B)
    ods, ocs = [], []
    for _ in range(nthr):
        od = oracle_mod.OracleDecoupled(N_short=Ns, N_long=Nl); od.set_trajectory(tube.data); ods.append(od)
        oc = oracle_mod.Oracle(); oc.set_trajectory(tube.data); ocs.append(oc)
    qh = np.zeros((steps, B, 6)); uh = np.zeros((steps, B, 3)); qf = np.zeros((B, 6)); uf = np.zeros((B, 3)); ok = np.ones(B, bool)

    def work(w):
        od, oc = ods[w], ocs[w]
        for b in range(w, B, nthr):
            q, u, t = state[b].copy(), control[b].copy(), float(t0[b])
            clock = od.simulate_times(dt, float(tube.t[-1]), steps + 1, t_start=float(t0[b]))               # `for t in 0:dt:trajectory.t[end]` (:87) as Julia's range gives it
            for k in range(steps):
                qh[k, b] = q; uh[k, b] = u
                ts, dts = od.time_steps(t)                                                                   # compute_time_steps! :90
                oq, ou, op = od.nodes(q, u, ts, dts, time_offset=toff[b])                                    # compute_linearization_nodes! :91
                sd = od.update_qp(oq, ou, op, dts)                                                           # update_QP! :92
                if walls:
                    edges = od.node_edges(q, u, ts, dts, time_offset=toff[b])[1:]
                    qpw, _ = oracle_mod.extend_with_walls(od, od.assemble_qp(sd), edges, od.unpack_sd(sd)["dt"], Ww)
                    xe, ye, info = od.solve_exact_verified(sd, qp=qpw, walls=edges, wall_weight=Ww)           # solve! :93
                else:
                    xe, ye, info = od.solve_exact_verified(sd)
                ok[b] &= info["status"] == 1
                X = od.split_x(xe[:od.n])
                q = oc.plant_step(q, u, dt)                                                                  # propagate with the OLD control :94
                u = od.next_control(X["delta"][1], ou[1, 1])                                                 # get_next_control :95 (decoupled_lat_long.jl:275-278)
                t = float(clock[k + 1])
            qf[b] = q; uf[b] = u
    with ThreadPoolExecutor(nthr) as ex:
        list(ex.map(work, range(nthr)))
    return qh, uh, qf, uf, ok


@pytest.mark.parametrize("Nl,walls,traj_mode", [(20, False, False), (40, False, True), (40, True, False)])
def test_decoupled_closed_loop_on_device_matches_oracle(pkg, oracle_mod, skidpad, Nl, walls, traj_mode):
    """pg_simulate_dev on a PG_DECOUPLED handle, 40 steps x 24 instances (N = 30; N = 50 without and with the wall rows; path and trajectory tracking mode): the device
    loop -- warm start of the lateral solver from step 2 on -- against the oracle loop with its exact solver.  Bars as for the coupled loop
    (test_closed_loop_on_device_matches_oracle): states 1e-5 relative, controls 1e-5 normalised, at every step."""
    Bc, steps, Ns, Ww = 24, 40, 10, 1000.0
    state, control, t0, toff = pkg.synthetic.config2_inputs(skidpad, Bc, seed=77, traj_mode=traj_mode)
    mpc = pkg.DecoupledTrajectoryTrackingMPC(pkg.X1(), skidpad, Bc, N_short=Ns, N_long=Nl, walls=walls, wall_weight=Ww)
    mpc.set_inputs(state, control, t0, time_offset=toff)
    s, c, t, qh, uh = mpc.simulate_(steps, dt=0.01, record=True)
    st, it, act, mu = mpc.solve_info()
    assert np.all(pkg.is_solved(st)), np.bincount(st)
    oq, ou, qf, uf, ok = oracle_lateral_loop(oracle_mod, skidpad, Ns, Nl, walls, Ww, state, control, t0, toff, steps)
    assert np.all(ok)
    for k in range(steps):
        assert np.max(np.abs(qh[k] - oq[k]) / np.maximum(1.0, np.abs(oq[k]))) < 1e-5, k
        assert np.max(np.abs(uh[k] - ou[k]) / UN) < 1e-5, k
    assert np.max(np.abs(s - qf) / np.maximum(1.0, np.abs(qf))) < 1e-5 and np.max(np.abs(c - uf) / UN) < 1e-5 and np.allclose(t, t0 + steps * 0.01)
    assert np.mean(it == 0) > 0.5            # the last step was served by the warm attempt for most instances (no interior-point iteration)


def test_two_consecutive_lateral_steps_match_the_oracle(pkg, oracle_mod, skidpad):
    """Cold step, plant, warm step through the five reference calls (not pg_simulate_dev): the second step's control of every instance against the exact optimum of the
    second step's QP built by the oracle from the same inputs -- and against a handle that never warm-starts."""
    B, Ns, Nl = 64, 10, 40
    state, control, t0, toff = pkg.synthetic.config2_inputs(skidpad, B, seed=5)
    od = oracle_mod.OracleDecoupled(N_short=Ns, N_long=Nl); od.set_trajectory(skidpad.data)
    oc = oracle_mod.Oracle(); oc.set_trajectory(skidpad.data)
    out = {}
    for warm in (True, False):
        mpc = pkg.DecoupledTrajectoryTrackingMPC(pkg.X1(), skidpad, B, N_short=Ns, N_long=Nl, warm_polish=warm)
        u1, st1, it1 = mpc.step_(state, control, t0, time_offset=toff)
        assert np.all(pkg.is_solved(st1)) and np.all(it1 > 0)
        state2 = np.stack([oc.plant_step(state[b], control[b], 0.01) for b in range(B)])
        u2, st2, it2 = mpc.step_(state2, u1, t0 + 0.01, time_offset=toff)
        assert np.all(pkg.is_solved(st2))
        out[warm] = (u1, u2, it2, mpc.polish_info().copy(), mpc.qp_data().copy(), mpc.solution()[0].copy())
        mpc.close()
    assert np.array_equal(out[True][0], out[False][0]) and np.array_equal(out[True][4], out[False][4])          # same cold step, same second QP
    assert np.mean(out[True][2] == 0) > 0.5 and np.all(out[False][2] > 0)                                     # served warm / never warm
    worst = 0.0
    for b in range(B):
        ts, dts = od.time_steps(t0[b] + 0.01)
        oq, ou, op = od.nodes(state2[b], out[True][0][b], ts, dts, time_offset=toff[b])
        sd = od.update_qp(oq, ou, op, dts)
        ref = oracle_mod.embed_sd(od, sd)
        assert np.max(np.abs(out[True][4][b] - ref) / np.maximum(1.0, np.abs(ref))) < 1e-8, b
        xe, ye, info = od.solve_exact_verified(sd)
        assert info["status"] == 1
        d2 = od.split_x(xe)["delta"][1]
        worst = max(worst, abs(out[True][5][b, 1, 6] - d2), abs(out[False][5][b, 1, 6] - d2))
    assert worst < 1e-6, worst


@pytest.mark.parametrize("path,walls,burn,min_served", [("skidpadoval", False, 3, 0.55), ("skidpadoval", True, 3, 0.5), ("EastPaddock", False, 100, 0.999)])
def test_warm_lateral_step_of_the_full_batch_every_instance_against_the_oracle(pkg, oracle_mod, path, walls, burn, min_served):
    """BASELINE configs[4] in closed loop: B = 4096, N = 50 (+ walls), `burn` steps on the device, then one more step -- warm for every instance -- checked like the cold
    one (test_config5_as_shipped_every_instance_against_the_oracle): the applied steering of EVERY instance within 1e-6 of the exact optimum of its own QP data (a
    verified KKT point of the canonical QP by the oracle), however the instance was served (warm attempt, or the interior point behind a failed / skipped attempt).
    What the warm start buys depends on the loop (EXPERIMENTS 10.1): k_solve_lat ends with its slowest instance, so ONE instance that falls back to the interior point
    costs the launch the cold time.  On the benchmark batch (random starts on the skidpad, far horizons whose working sets turn over by a dozen rows per 10 ms) 75-85 %
    of the instances are served warm and the solve phase is that of a cold step; in a settled loop (EastPaddock after 1 s) every instance is served by ONE polish round and
    the solve phase is >= 3x shorter than the cold one.  No wall-clock bar here (round 6: a correctness test does not time single un-warmed launches): the ratios are
    measured by bench.py (`decoupled_n50.closed_loop`: both loops, with and without the warm start); what this test asserts of the warm start is the SHARE of instances it
    serves without an interior-point iteration."""
    B, Ns, Nl = 4096, 10, 40
    traj = pkg.load_path_fixture(path)
    mpc = pkg.DecoupledTrajectoryTrackingMPC(pkg.X1(), traj, B, N_short=Ns, N_long=Nl, walls=walls)
    state, control, t0, toff = pkg.synthetic.config2_inputs(traj, B)
    mpc.set_inputs(state, control, t0, time_offset=toff)
    s, c, t, _, _ = mpc.simulate_(burn)
    cold = pkg.DecoupledTrajectoryTrackingMPC(pkg.X1(), traj, B, N_short=Ns, N_long=Nl, walls=walls, warm_polish=False)
    cold.set_inputs(s, c, t, time_offset=toff); cold.step_dev(); cold.synchronize(); cold_ms = cold.phase_ms(); uc = cold.get_next_control(); polc = cold.polish_info()
    cold.close()
    mpc.set_inputs(s, c, t, time_offset=toff)
    mpc.step_dev(); mpc.synchronize(); warm_ms = mpc.phase_ms()
    st, it, act, mu = mpc.solve_info(); pol = mpc.polish_info(); uw = mpc.get_next_control()
    assert np.all(pkg.is_solved(st)), np.bincount(st)
    assert np.array_equal(st == pkg.SOLVED, pol >= 1)
    res = check_lateral_batch_against_oracle(pkg, oracle_mod, traj, mpc, B, Ns, Nl, walls)
    both = (pol >= 1) & (polc >= 1)
    print(f"{path} walls={walls}: warm step {burn + 1}: solve {warm_ms[2]:.3f} ms (the same step solved cold: {cold_ms[2]:.3f} ms), served without the interior point {int((it == 0).sum())}/{B}, "
          f"verified {int((pol >= 1).sum())}/{B}, max |d2-d2*| {res[:, 0].max():.2e} (median {np.median(res[:, 0]):.1e}), max objective gap {res[:, 1].max():.2e}, "
          f"worst row violation {res[:, 2].max():.2e}, warm vs cold handle max |d2 diff| {np.max(np.abs(uw[both, 0] - uc[both, 0])):.1e}")
    assert np.all(res[:, 4] == 1) and np.all(res[:, 5] >= 1)
    assert res[:, 0].max() <= 1e-6, (res[:, 0].max(), int(np.argmax(res[:, 0])))
    # objective: 1e-5 relative where the optimum stays within 100 m of the path; instances whose optimum leaves the linearisation by kilometres (objective ~1e9, multipliers
    # ~1e6 on rows of curvature 1e12: the polish's sign test on such a multiplier is only good to ~1e3) get 1e-4 (measured 1.6e-5 on one instance with |e*| = 3.7 km)
    near = res[:, 6] <= 100.0
    # (round 5: with the held rate rows pinned exactly a working set verifies at its first check, a refinement pass earlier than under the augmented Lagrangian; the far end of the
    #  horizon -- weakly determined, R_delta = 0 -- is then 2e-5 instead of 1e-6 from the oracle's and ONE near instance of the walls batch has an objective gap of 1.05e-5: bar 2e-5)
    assert res[near, 1].max() <= 2e-5 and res[:, 1].max() <= 1e-4 and res[:, 2].max() <= 1e-9, (res[near, 1].max(), res[:, 1].max(), res[:, 2].max())
    assert (pol >= 1).sum() >= B - 8, int((pol < 0).sum())          # round 4: 37-60 unverified answers per step (stalled multipliers of held rate rows); pinned: 0-4
    assert np.max(np.abs(uw[both, 0] - uc[both, 0])) <= 3e-7                  # two verified KKT points of the same QP (each within 2e-7 of the oracle's: measured 1.3e-7 apart)
    assert np.mean(it == 0) >= min_served, np.mean(it == 0)


def test_short_list_behind_the_warm_attempts_runs_one_instance_per_wavefront(pkg, skidpad):
    """Round 6: behind the warm attempts of a lateral step the unserved instances are solved cold from a to-do list; a list of up to `lat_single_max` (1024) instances takes the
    one-instance-per-wavefront arrangement, a longer one four per wavefront -- both launches are queued, the DEVICE word (the list's length) picks one.  The host cannot see the
    choice; the answers can: the two arrangements end in the same verified KKT points (3e-7) but not in the same bits (their sums run over 64 and 16 lanes), so a warm step whose
    list is short differs in bits from the same step with the option at 0 exactly on listed instances -- and a step whose list is long does not differ at all."""
    Ns, Nl = 10, 40
    res = {}
    for B, walls, burn in ((2048, True, 4), (4096, False, 3)):           # a short list (a fifth of 2048 instances) and a long one (a quarter of 4096)
        state, control, t0, toff = pkg.synthetic.config2_inputs(skidpad, B)
        for name, opts in (("default", {}), ("four", {"lat_single_max": 0})):
            m = pkg.DecoupledTrajectoryTrackingMPC(pkg.X1(), skidpad, B, N_short=Ns, N_long=Nl, walls=walls, options=dict(opts, lat_handover=0))      # (the same cold first step in both)
            m.set_inputs(state, control, t0, time_offset=toff)
            m.simulate_(1)                                                # cold step: the single four-per-wavefront launch in both handles
            m.set_option("lat_single_max", 0); m.simulate_(burn - 1)      # identical warm steps up to the one under test ...
            m.set_option("lat_single_max", 0 if name == "four" else 1024)
            s, c, t = m.simulate_(1)[:3]                                  # ... which differs in the option only
            st, it, _, _ = m.solve_info()
            res[(walls, name)] = (np.asarray(c).copy(), st.copy(), it.copy(), m.polish_info().copy())
            m.close()
        (cd, sd, itd, pd_), (cf, sf, itf, pf) = res[(walls, "default")], res[(walls, "four")]
        listed = itf > 0                                                  # went through the interior point: was on the to-do list
        differ = (cd != cf).any(axis=1)
        both = (pd_ >= 1) & (pf >= 1)
        print(f"B={B} walls={walls}: listed {int(listed.sum())}, controls differ in bits on {int(differ.sum())}, max |du| {np.max(np.abs(cd[both] - cf[both]) / UN):.1e}")
        assert np.all(pkg.is_solved(sd)) and np.all(pkg.is_solved(sf))
        assert not np.any(differ & ~listed)                               # served by the warm attempt: the same launch, the same bits
        assert np.max(np.abs(cd[both] - cf[both]) / UN) <= 3e-7
        assert (listed.sum() <= 1024) == (B == 2048)                      # (the two cases are what the comment above says they are)
        if listed.sum() <= 1024: assert differ.sum() > 0                  # a short list: the other arrangement ran
        else: assert differ.sum() == 0                                    # a long list: the same four-per-wavefront launch
