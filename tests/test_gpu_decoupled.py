"""GPU parity of the decoupled (lateral) formulation (decoupled_lat_long.jl) against the CPU oracle, N = 30 and N = 50 (BASELINE config 5)."""
import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("Ns,Nl", [(10, 20), (10, 40)])
def test_decoupled_matches_oracle(pkg, oracle_mod, skidpad, Ns, Nl):
    B = 64
    mpc = pkg.DecoupledTrajectoryTrackingMPC(pkg.X1(), skidpad, B, N_short=Ns, N_long=Nl, polish=True)
    assert np.array_equal(mpc.u_normalization, [1.0, 1.0])
    orc = oracle_mod.OracleDecoupled(N_short=Ns, N_long=Nl); orc.set_trajectory(skidpad.data)
    assert (orc.n, orc.m) == ((245, 455) if Nl == 20 else (405, 755))
    state, control, t0, toff = pkg.synthetic.config2_inputs(skidpad, B, seed=31, traj_mode=(Nl == 20))
    u, status, iters = mpc.step_(state, control, t0, time_offset=toff)
    assert np.all(pkg.is_solved(status)), status      # (PG_SOLVED_UNVERIFIED: instances whose polish did not verify keep the interior-point iterate; compared below like the rest)
    qs, us, ps = mpc.nodes(); qp = mpc.qp_data(); x, sg = mpc.solution(); st, it, act, mu = mpc.solve_info(); lam = mpc.multipliers()
    worst = 0.0; lam_err = 0.0; n_lam = 0
    for b in range(B):
        ts, dt = orc.time_steps(t0[b])
        oq, ou, op = orc.nodes(state[b], control[b], ts, dt, time_offset=toff[b])
        assert np.max(np.abs(qs[b, :, 2:] - oq)) < 1e-9 and np.max(np.abs(us[b] - ou)) <= 1e-9 * max(1, np.max(np.abs(ou)))
        assert np.max(np.abs(qs[b, :, 1] - op[:, 0])) < 1e-9 and np.max(np.abs(ps[b, :, 1] - op[:, 1])) < 1e-12      # Ux parameter, kappa
        sd = orc.update_qp(oq, ou, op, dt)
        ref = oracle_mod.embed_sd(orc, sd)
        assert np.max(np.abs(qp[b] - ref) / np.maximum(1.0, np.abs(ref))) < 1e-8, b
        xe, ye, info = orc.solve_exact(sd); X = orc.split_x(xe)
        assert info["status"] == 1
        worst = max(worst, abs(x[b, 1, 6] - X["delta"][1]))
        # Far-horizon steering is only weakly determined in the lateral QP (R_delta = 0, saturated nodes): compare the whole trajectory through
        # an optimality certificate on the canonical QP instead of entry by entry: same objective value, all rows satisfied.
        qpc = orc.assemble_qp(sd)
        xg = np.concatenate([x[b, :, 2:6].ravel(), x[b, :, 6], sg[b, :, :2].ravel(), np.diff(x[b, :, 6])])
        Ac = sp.csc_matrix((qpc["Ax"], qpc["Ai"], qpc["Ap"]), shape=(orc.m, orc.n))
        obj = lambda v: 0.5 * np.dot(qpc["Pd"] * v, v) + np.dot(qpc["q"], v)
        assert abs(obj(xg) - obj(xe)) <= 1e-6 * (1.0 + abs(obj(xe))), (b, obj(xg), obj(xe))
        Axg = Ac @ xg
        assert max(np.max(qpc["l"] - Axg), np.max(Axg - qpc["u"])) < 1e-8
        assert np.max(np.abs(x[b, :, 6] - X["delta"])) < 1e-3 and np.max(np.abs(x[b, :, 2:6] - X["q"])) < 1e-3
        # inert slots of the embedding stay put
        assert np.all(x[b, :, 0] == 0) and np.all(x[b, :, 7] == 0) and np.max(np.abs(x[b, :, 1] - 8.0)) < 1e-12
        uo = orc.next_control(X["delta"][1], ou[1, 1])                       # decoupled_lat_long.jl:275-278
        assert abs(u[b, 0] - uo[0]) < 1e-6 and np.max(np.abs(u[b, 1:] - uo[1:])) <= 1e-9 * max(1.0, np.max(np.abs(uo)))
        assert pkg.decoupled_canonical_active_set(orc.N, Ns, act[b], lam=lam[b]) == oracle_mod.active_set(qpc, xe, ye, tol=1e-6), b
        # multipliers of the held steering rows of a VERIFIED instance against the oracle's duals: the rate rows (bits 12 / 13) are eliminated exactly in k_solve_lat's polish
        # and their multipliers read off stationarity (round 5), the steering-bound rows (3 / 4) go through the augmented Lagrangian
        if st[b] == pkg.SOLVED:
            r_7 = 2 * orc.N + orc.N + 4 + 1 + 4 * Ns + 4 * (orc.N - Ns)
            for k in range(orc.N):
                for j, row in ((3, r_7 + 8 * k), (4, r_7 + 8 * k + 1), (12, r_7 + 8 * k + 6), (13, r_7 + 8 * k + 7)):
                    if (int(act[b, k]) >> j) & 1 and lam[b, k, j] > 1e-6:
                        lam_err = max(lam_err, abs(lam[b, k, j] - abs(ye[row])) / max(1.0, abs(ye[row]))); n_lam += j >= 12
    assert worst < 1e-6, worst
    assert n_lam >= 20 and lam_err <= 1e-5, (n_lam, lam_err)      # (measured: see the printed line)
    print(f"N = {orc.N}: {n_lam} held rate rows compared, worst |lambda - lambda*| / max(1, lambda*) over the steering rows {lam_err:.1e}")


def test_decoupled_refuses_hji(pkg, skidpad):
    mpc = pkg.DecoupledTrajectoryTrackingMPC(pkg.X1(), skidpad, 4)
    knots, V, g = pkg.synthetic.hji_grid(dims=(3, 3, 3, 3, 3, 3, 3))
    with pytest.raises(pkg.PigeonError):
        mpc.set_hji_cache(knots, V, g)


def test_walls_extension_matches_oracle_qp_with_wall_rows(pkg, oracle_mod, skidpad):
    """BUILD-DEFINED extension (BASELINE config 5 "both_walls"; the reference snapshot has no wall constraint): soft rows
    e_t <= edge_L(s_t) + sw_t, e_t >= edge_R(s_t) - sw_t, sw_t >= 0 with cost wall_weight dt_t sw_t at nodes 2..N+1 of the lateral QP.  Checked
    against the oracle's canonical lateral QP extended in numpy (N slack columns, 3N rows) and solved by the oracle's sparse interior point.
    The left wall sits at e = -0.05 m and every instance starts at e0 in [-0.5, -0.1]: the tracking cost pulls e to 0, so the wall binds."""
    Ns, Nl, B, Ww = 10, 40, 48, 1000.0
    t = skidpad
    tube = pkg.TrajectoryTube(t.t, t.s, t.V, t.A, t.E, t.N, t.psi, t.kappa, edge_L=np.full(len(t), -0.05), edge_R=np.full(len(t), -4.0))
    rng = np.random.default_rng(5)
    s = rng.uniform(5.0, tube.s[-1] - 80.0, B)
    E, N, psi, kappa, V, tt = pkg.synthetic.path_pose(tube, s)
    e = rng.uniform(-0.5, -0.1, B)
    state = np.stack([E - e * np.cos(psi), N - e * np.sin(psi), psi + rng.uniform(-0.05, 0.05, B), V * rng.uniform(0.95, 1.05, B), rng.uniform(-0.1, 0.1, B),
                      kappa * V + rng.uniform(-0.02, 0.02, B)], axis=1)
    control = np.stack([rng.uniform(-0.02, 0.02, B), np.zeros(B), rng.uniform(0, 300.0, B)], axis=1)
    t0 = tt + rng.uniform(-0.1, 0.1, B); toff = np.zeros(B)
    mpc = pkg.DecoupledTrajectoryTrackingMPC(pkg.X1(), tube, B, N_short=Ns, N_long=Nl, walls=True, wall_weight=Ww, polish=True)
    free = pkg.DecoupledTrajectoryTrackingMPC(pkg.X1(), tube, B, N_short=Ns, N_long=Nl)
    u, status, iters = mpc.step_(state, control, t0, time_offset=toff)
    uf, stf, _ = free.step_(state, control, t0, time_offset=toff)
    assert np.all(pkg.is_solved(status)), np.bincount(status)
    x, sg = mpc.solution(); st, it, act, mu = mpc.solve_info(); edges = mpc.wall_edges(); xf, _ = free.solution()
    orc = oracle_mod.OracleDecoupled(N_short=Ns, N_long=Nl); orc.set_trajectory(tube.data)
    n, m, Nh = orc.n, orc.m, orc.N
    n_active = 0; worst = 0.0
    for b in range(B):
        ts, dt = orc.time_steps(t0[b])
        oq, ou, op = orc.nodes(state[b], control[b], ts, dt, time_offset=toff[b])
        oe = orc.node_edges(state[b], control[b], ts, dt, time_offset=toff[b])
        assert np.max(np.abs(edges[b] - oe[1:])) < 1e-12
        sd = orc.update_qp(oq, ou, op, dt)
        qpc = orc.assemble_qp(sd)
        # extend: columns n..n+N-1 = sw_k; rows m+3k: e - sw <= edge_L, m+3k+1: e + sw >= edge_R, m+3k+2: sw >= 0; e = component 3 of q = (Uy, r, dpsi, e)
        A = sp.csc_matrix((qpc["Ax"], qpc["Ai"], qpc["Ap"]), shape=(m, n))
        W = sp.lil_matrix((3 * Nh, n + Nh)); lw = np.full(3 * Nh, -1e20); uw = np.full(3 * Nh, 1e20)
        for k in range(Nh):
            col = 4 * (k + 1) + 3
            W[3 * k, col] = 1.0; W[3 * k, n + k] = -1.0; uw[3 * k] = oe[k + 1, 0]
            W[3 * k + 1, col] = 1.0; W[3 * k + 1, n + k] = 1.0; lw[3 * k + 1] = oe[k + 1, 1]
            W[3 * k + 2, n + k] = 1.0; lw[3 * k + 2] = 0.0
        Aw = sp.vstack([sp.hstack([A, sp.csc_matrix((m, Nh))]), W]).tocsc(); Aw.sort_indices()
        qpw = dict(Pd=np.concatenate([qpc["Pd"], np.zeros(Nh)]), q=np.concatenate([qpc["q"], Ww * dt]), Ap=Aw.indptr, Ai=Aw.indices, Ax=Aw.data,
                   l=np.concatenate([qpc["l"], lw]), u=np.concatenate([qpc["u"], uw]))
        xe, ye, info = oracle_mod.solve_exact_generic(qpw)
        assert info["status"] == 1, (b, info)
        X = orc.split_x(xe[:n])
        worst = max(worst, abs(x[b, 1, 6] - X["delta"][1]))
        xg = np.concatenate([x[b, :, 2:6].ravel(), x[b, :, 6], sg[b, :, :2].ravel(), np.diff(x[b, :, 6]), sg[b, :, 2]])
        obj = lambda v: 0.5 * np.dot(qpw["Pd"] * v, v) + np.dot(qpw["q"], v)
        assert abs(obj(xg) - obj(xe)) <= 1e-6 * (1.0 + abs(obj(xe))), (b, obj(xg), obj(xe))
        Axg = Aw @ xg
        assert max(np.max(qpw["l"] - Axg), np.max(Axg - qpw["u"])) < 1e-8
        # active sets: the far-horizon steering is not unique (R_delta = 0) and where a wall starts or stops binding the multipliers are not unique
        # either (sw = 0 together with a tight wall row), so the index lists are compared as a sandwich: every strongly active row of the oracle is
        # active on the GPU, and every GPU-active row is tight at the GPU's optimum (whose optimality the objective/feasibility certificate above shows)
        G = set(pkg.decoupled_canonical_active_set(Nh, Ns, act[b], walls=True))
        strong = set(oracle_mod.active_set(qpw, xe, ye, tol=1e-3))
        tight = {+(i + 1) for i in range(len(Axg)) if qpw["u"][i] - Axg[i] < 1e-6} | {-(i + 1) for i in range(len(Axg)) if Axg[i] - qpw["l"][i] < 1e-6}
        assert strong <= G <= tight, (b, sorted(strong - G, key=abs)[:5], sorted(G - tight, key=abs)[:5])
        n_active += sum(1 for k in range(Nh) if int(act[b][k]) & 1)
    assert worst < 2e-5, worst          # (N = 50 with binding walls: a third of the instances stop at the rounding floor mu ~ 1e-10, see EXPERIMENTS.md 4.3)
    assert n_active > 5 * B                                                            # the wall really binds
    assert np.max(xf[:, 1:, 5]) > -0.04                                                # ... and without it the optimum crosses e = -0.05
    with pytest.raises(pkg.PigeonError):
        pkg.BatchedTrajectoryTrackingMPC(tube, 4, walls=True)                         # coupled + walls is refused


# ------------------------------------------------------------------------------------------------------------------
# BASELINE configs[4] at FULL size, in the solver configuration the library ships (pg_default_config_decoupled): every instance against the oracle

def check_lateral_batch_against_oracle(pkg, oracle_mod, tube, mpc, B, Ns, Nl, walls, Ww=1000.0, want_more=False):
    """Every instance of the batch `mpc` just solved: exact optimum of ITS OWN QP data by the oracle (threaded), as a VERIFIED KKT point of the canonical QP
    (OracleDecoupled.solve_exact_verified).  Returns per instance (|delta_2 - delta_2*|, relative objective gap, worst row violation relative to 1 + |A x|_inf, max |delta - delta*| over the
    horizon, oracle status, oracle polish rounds (>= 1: verified), max |e*| over the horizon, max sigma*)."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    qp = mpc.qp_data(); x, sg = mpc.solution()
    edges = mpc.wall_edges() if walls else None
    nthr = min(16, len(os.sched_getaffinity(0)))
    orcs = []
    for _ in range(nthr):
        o = oracle_mod.OracleDecoupled(N_short=Ns, N_long=Nl); o.set_trajectory(tube.data); orcs.append(o)

    def work(w):
        o = orcs[w]; out = []
        for b in range(w, B, nthr):
            sd = oracle_mod.unembed_qp(o, qp[b]); qpc = o.assemble_qp(sd)
            xg = np.concatenate([x[b, :, 2:6].ravel(), x[b, :, 6], sg[b, :, :2].ravel(), np.diff(x[b, :, 6])])
            if walls:
                S = o.unpack_sd(sd)
                qpw, Ac = oracle_mod.extend_with_walls(o, qpc, edges[b], S["dt"], Ww)
                xe, ye, info = o.solve_exact_verified(sd, qp=qpw, walls=edges[b], wall_weight=Ww)
                xg = np.concatenate([xg, sg[b, :, 2]])
            else:
                qpw = qpc; Ac = sp.csc_matrix((qpc["Ax"], qpc["Ai"], qpc["Ap"]), shape=(o.m, o.n))
                xe, ye, info = o.solve_exact_verified(sd)
            X = o.split_x(xe[:o.n])
            obj = lambda v: 0.5 * np.dot(qpw["Pd"] * v, v) + np.dot(qpw["q"], v)
            Axg = Ac @ xg
            out.append((abs(x[b, 1, 6] - X["delta"][1]), (obj(xg) - obj(xe)) / (1.0 + abs(obj(xe))), max(np.max(qpw["l"] - Axg), np.max(Axg - qpw["u"])) / (1.0 + np.max(np.abs(Axg))),
                        float(np.max(np.abs(x[b, :, 6] - X["delta"]))), info["status"], info["polished"], float(np.max(np.abs(X["q"][:, 3]))), float(np.max(X["sigma"]))))
        return out
    with ThreadPoolExecutor(nthr) as ex:
        parts = list(ex.map(work, range(nthr)))
    res = np.zeros((B, 8))
    for w, part in enumerate(parts):
        res[w:B:nthr] = np.array(part)
    return res


@pytest.mark.parametrize("walls", [False, True])
def test_config5_as_shipped_every_instance_against_the_oracle(pkg, oracle_mod, skidpad, walls):
    """BASELINE configs[4]: B = 4096 lateral MPCs, N = 50 (N_short 10 + N_long 40), the bench's own batch (config2_inputs, seed 12345), solved with the
    DEFAULT solver configuration of the library (pg_default_config_decoupled: k_solve_lat, interior point handed over to its active-set polish) -- with the
    wall rows (the bench's configs[4] line) and without (the reference's lateral QP as it stands).  The oracle's answer is a VERIFIED KKT point of the canonical
    QP for every one of the 4096 instances (solve_exact_verified; a third of these QPs have optima that leave the linearisation by more than 10 m, a few by
    kilometres -- saturated steering on an open-loop unstable 8 s horizon -- and the sparse interior point of oracle/qp.hpp alone fails on ~1 % of them).
    Bar: the applied steering delta_2 within 1e-6 of the exact optimum of the same QP data for EVERY instance (measured: 1.3e-7 / 8e-8, median 2e-14);
    the objective within 1e-5 relative (measured 1e-7 / 4e-6: the far horizon is weakly determined, R_delta = 0); every row of the canonical QP satisfied to 1e-9
    of the size of A x.  With the interior point alone (polish=False, the round-2 default) 5 / 48 instances sit 1e-6 .. 1e-4 away: asserted below as the reason
    the default changed."""
    B, Ns, Nl = 4096, 10, 40
    mpc = pkg.DecoupledTrajectoryTrackingMPC(pkg.X1(), skidpad, B, N_short=Ns, N_long=Nl, walls=walls)
    state, control, t0, toff = pkg.synthetic.config2_inputs(skidpad, B)
    u, status, iters = mpc.step_(state, control, t0, time_offset=toff)
    assert np.all(pkg.is_solved(status)), np.bincount(status)
    assert np.array_equal(status == pkg.SOLVED, mpc.polish_info() >= 1)             # PG_SOLVED = a verified KKT point, PG_SOLVED_UNVERIFIED = the interior-point iterate
    res = check_lateral_batch_against_oracle(pkg, oracle_mod, skidpad, mpc, B, Ns, Nl, walls)
    pol = mpc.polish_info()
    print(f"walls={walls}: max |d2-d2*| {res[:, 0].max():.2e} (median {np.median(res[:, 0]):.1e}), max objective gap {res[:, 1].max():.2e}, worst row violation {res[:, 2].max():.2e}, "
          f"max |delta-delta*| over the horizon {res[:, 3].max():.2e}, iterations mean {iters.mean():.1f} max {iters.max()}, verified by the polish {int((pol >= 1).sum())}/{B}")
    assert np.all(res[:, 4] == 1) and np.all(res[:, 5] >= 1)                      # the oracle's side: a verified KKT point for every instance
    assert res[:, 0].max() <= 1e-6, (res[:, 0].max(), int(np.argmax(res[:, 0])))
    assert res[:, 1].max() <= 1e-5 and res[:, 2].max() <= 1e-9, (res[:, 1].max(), res[:, 2].max())
    assert (pol >= 1).sum() >= B - 8, int((pol < 0).sum())         # (round 4: 53 / 54 unverified -- multipliers of held steering-rate rows stalled under the penalty; round 5 pins them exactly: 0 / 4)
    # round 6: the default cold launch of this batch is the two-launch straggler hand-over (k_solve_lat<.., 16, 1> -> k_solve_lat<.., 64, 2>): what has just been held
    # against the oracle, every instance of it, IS that path -- the library's own counter says so
    assert mpc.get_option("stat_lat_handover_solves") == 1 and mpc.get_option("lat_handover") == 1
    # ... and the single launch of round 5 (option "lat_handover" = 0) gives the same answers: both end in verified KKT points of the same QP (or, for the handful that
    # verify nowhere, in interior-point iterates at 1e-12).  Iteration counts agree except where the order of a sum decided a step rule (one instance per wavefront sums over
    # 64 lanes, four per wavefront over 16)
    one = pkg.DecoupledTrajectoryTrackingMPC(pkg.X1(), skidpad, B, N_short=Ns, N_long=Nl, walls=walls, options={"lat_handover": 0})
    u1, status1, iters1 = one.step_(state, control, t0, time_offset=toff)
    assert one.get_option("stat_lat_handover_solves") == 0 and np.all(pkg.is_solved(status1))
    both = (status == pkg.SOLVED) & (status1 == pkg.SOLVED)
    print(f"walls={walls}: hand-over vs single launch: status differs on {int((status != status1).sum())}, iterations on {int((iters != iters1).sum())}, max |d2 - d2'| {np.max(np.abs(u[:, 0] - u1[:, 0])):.1e}")
    assert both.sum() >= B - 12 and np.max(np.abs(u[both, 0] - u1[both, 0])) <= 3e-7 and np.max(np.abs(u[:, 0] - u1[:, 0])) <= 1e-6 and (iters != iters1).sum() <= B // 20
    one.close()
    mpc.close()
    # the interior point alone: accurate for all but a handful -- which is why it is not the default
    ipm = pkg.DecoupledTrajectoryTrackingMPC(pkg.X1(), skidpad, B, N_short=Ns, N_long=Nl, walls=walls, polish=False)
    u2, status2, _ = ipm.step_(state, control, t0, time_offset=toff)
    assert np.all(status2 == pkg.SOLVED)
    d = np.abs(u2[:, 0] - u[:, 0])
    print(f"walls={walls}: interior point alone vs default: {int((d > 1e-6).sum())} instances differ by more than 1e-6 in delta_2 (max {d.max():.1e}, 99.9th percentile {np.percentile(d, 99.9):.1e})")
    assert np.percentile(d, 95) < 1e-6 and d.max() < 1e-3
    ipm.close()


@pytest.mark.gpu
@pytest.mark.parametrize("B,walls,path", [(1, False, "skidpadoval"), (5, True, "skidpadoval"), (1021, True, "EastPaddock"), (130, False, "vail")])
def test_lateral_kernel_on_ragged_batches_matches_the_embedding(pkg, B, walls, path):
    """k_solve_lat against the embedding of the same QP in k_solve (option "lateral_solver" = 2, one wavefront per instance) on batch sizes that leave the last wavefront
    ragged and on other paths than the benchmark's: two different kernels, the same verified KKT point wherever both verify.  Round 6: cold batches of up to 1024
    instances take k_solve_lat's ONE-instance-per-wavefront instantiation (option "lat_single_max"; the library's counter says so); with the option at 0 the same batch runs
    four per wavefront with the row state in the workspace, ragged last wavefront included -- both are held against the embedding, and against each other."""
    traj = pkg.load_path_fixture(path)
    state, control, t0, toff = pkg.synthetic.config2_inputs(traj, B, seed=3)
    out = {}
    for lat, opts in (("1", {"lateral_solver": 1}), ("4", {"lateral_solver": 1, "lat_single_max": 0}), ("0", {"lateral_solver": 2})):
        mpc = pkg.DecoupledTrajectoryTrackingMPC(pkg.X1(), traj, B, N_short=10, N_long=40, walls=walls, options=opts)
        assert mpc.get_option("lateral_solver_in_use") == (2 if lat == "0" else 1)
        u, status, iters = mpc.step_(state, control, t0, time_offset=toff)
        assert mpc.get_option("stat_lat_one_per_wavefront_solves") == (1 if lat == "1" else 0)
        x, sg = mpc.solution()
        out[lat] = (u.copy(), status.copy(), x.copy(), mpc.polish_info().copy())
        mpc.close()
    (u4, s4, x4, p4) = out["4"]
    (ua, sa, xa, pa), (ue, se, xe, pe) = out["1"], out["0"]
    assert np.all(pkg.is_solved(s4))
    b14 = (pa >= 1) & (p4 >= 1)
    assert b14.sum() >= max(1, int(0.9 * B)) and np.max(np.abs(xa[b14, 1, 6] - x4[b14, 1, 6])) <= 3e-7      # one per wavefront against four per wavefront: two verified KKT points of the same QP
    assert np.all(pkg.is_solved(sa)), np.bincount(sa)
    both = (pa >= 1) & (pe >= 1)
    assert both.sum() >= max(1, int(0.8 * B)), (int(both.sum()), B)
    assert np.max(np.abs(xa[both, 1, 6] - xe[both, 1, 6])) <= 1e-7                    # applied steering: both are verified KKT points of the same QP data
    assert np.max(np.abs(ua[both] - ue[both])) <= 1e-6 * max(1.0, np.max(np.abs(ue[both])))
    solved_e = pkg.is_solved(se)
    assert np.max(np.abs(xa[solved_e, 1, 6] - xe[solved_e, 1, 6])) <= 1e-5            # interior-point iterates of either kernel where a polish did not verify


@pytest.mark.gpu
@pytest.mark.parametrize("B,walls,Nl,path", [(1025, True, 40, "skidpadoval"), (1538, False, 40, "vail"), (2051, True, 40, "EastPaddock"), (3001, False, 20, "skidpadoval"), (1300, True, 15, "skidpadoval")])
def test_handover_on_ragged_batches_and_other_horizons_matches_the_single_launch(pkg, B, walls, Nl, path):
    """Round 6: the straggler hand-over (first launch four instances per wavefront -> resuming launch one per wavefront) on batch sizes that are not multiples of four
    (ragged last wavefront of the first launch: its empty lane groups must neither be counted as finished nor filed), on other paths, and on horizons of 20 and 30 intervals
    (the hand-over needs the row state in the workspace: N > 16 since the register instantiation for 17..32 was retired; k_solve_lat is the default solver beyond 20 intervals).  Held against the
    single launch of round 5 (options lat_handover = 0, lat_single_max = 0): every instance solved by both, the same verified KKT point wherever both verify.  A second cold step on the
    same handle returns the same BITS under the default stopping rule (a trip count) and in the single launch; under the count rule (lat_hand_target) the same point to 1e-6."""
    traj = pkg.load_path_fixture(path)
    state, control, t0, toff = pkg.synthetic.config2_inputs(traj, B, seed=11)
    out = {}
    for name, opts in (("hand", {}), ("count", {"lat_hand_target": max(64, B // 3)}), ("one", {"lat_handover": 0, "lat_single_max": 0})):
        mpc = pkg.DecoupledTrajectoryTrackingMPC(pkg.X1(), traj, B, N_short=10, N_long=Nl, walls=walls, options=opts)
        u, status, iters = mpc.step_(state, control, t0, time_offset=toff)
        used = mpc.get_option("stat_lat_handover_solves")
        x, sg = mpc.solution()
        # a second cold step on the same handle (reset: nothing of the first hand-over may be left in the counters or the list)
        mpc.reset(); u2, status2, _ = mpc.step_(state, control, t0, time_offset=toff)
        # the default rule (a fixed number of trips) and the single launch depend on the data only: the same bits.  The count rule (option lat_hand_target) stops a wavefront
        # when it SEES few enough unfinished instances -- timing -- and resumes an instance a trip earlier or later: the same KKT point to 1e-7, not the same bits
        if name != "count": assert np.array_equal(status, status2) and np.array_equal(u, u2), name
        else: assert np.max(np.abs(u[:, 0] - u2[:, 0])) <= 1e-6
        out[name] = (u.copy(), status.copy(), iters.copy(), mpc.polish_info().copy(), used)
        mpc.close()
    (uh, sh, ih, ph, used_h), (uo, so, io, po, used_o) = out["hand"], out["one"]
    uc, sc, _, pc, used_c = out["count"]
    assert used_c == used_h and np.all(pkg.is_solved(sc))
    bc = (pc >= 1) & (ph >= 1)
    assert np.max(np.abs(uc[bc, 0] - uh[bc, 0])) <= 3e-7
    assert used_o == 0
    assert used_h == (1 if 10 + Nl > 16 else 0), used_h
    assert np.all(pkg.is_solved(sh)) and np.all(pkg.is_solved(so)), (np.bincount(sh), np.bincount(so))
    both = (ph >= 1) & (po >= 1)
    print(f"B={B} walls={walls} N={10 + Nl} {path}: verified {int((ph >= 1).sum())} / {int((po >= 1).sum())} of {B}, max |u - u'| {np.max(np.abs(uh[both] - uo[both])):.1e}, iterations differ on {int((ih != io).sum())}")
    assert both.sum() >= int(0.9 * B) and abs(int((ph >= 1).sum()) - int((po >= 1).sum())) <= max(4, B // 200)      # (vail: 96 % verify -- in either arrangement)
    assert np.max(np.abs(uh[both, 0] - uo[both, 0])) <= 3e-7
    assert np.max(np.abs(uh[:, 0] - uo[:, 0])) <= 1e-5
