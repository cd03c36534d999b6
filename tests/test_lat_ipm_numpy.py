"""The oracle's second exact method for the lateral formulation (oracle/lat_ipm_numpy.py: stage-structured interior point in numpy) and the verified solve built on
it (OracleDecoupled.solve_exact_verified): agreement with the sparse interior point of oracle/qp.hpp where both work, verification on the QPs where only one does."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def lateral_qps(pkg, oracle_mod, skidpad):
    """Every 64th instance of the BASELINE configs[4] batch (N = 50) + four instances on which the sparse interior point alone fails (found by the round-3 sweep)."""
    Ns, Nl, B = 10, 40, 4096
    state, control, t0, toff = pkg.synthetic.config2_inputs(skidpad, B)
    o = oracle_mod.OracleDecoupled(N_short=Ns, N_long=Nl); o.set_trajectory(skidpad.data)
    idx = sorted(set(range(0, B, 64)) | {528, 1024, 2192, 2224})
    sds = []
    for b in idx:
        ts, dt = o.time_steps(t0[b]); q, u, p = o.nodes(state[b], control[b], ts, dt, time_offset=toff[b]); sds.append(o.update_qp(q, u, p, dt))
    return o, idx, sds


def test_stage_interior_point_matches_the_sparse_one(oracle_mod, lateral_qps):
    from oracle import lat_ipm_numpy as lp
    o, idx, sds = lateral_qps
    worst = 0.0; n = 0
    for b, sd in zip(idx, sds):
        xe, ye, info = o.solve_exact(sd)
        if not (info["status"] == 1 and info["polished"] >= 1):
            continue
        D = lp.stage_data(o.unpack_sd(sd), o.cp); r = lp.solve(D)
        assert r["status"] == 1
        X = o.split_x(xe); n += 1
        worst = max(worst, abs(r["x"][1, 4] - X["delta"][1]))
        assert np.max(np.abs(r["t"])) < 1e6 and np.all(r["t"] > 0) and np.all(r["lam"] > 0)
    assert n >= 60 and worst < 1e-5, (n, worst)          # an interior-point iterate: sqrt(mu) on nearly degenerate rows (the polish is what makes it exact)


def test_verified_solve_covers_every_instance(oracle_mod, lateral_qps):
    o, idx, sds = lateral_qps
    methods = {}
    for b, sd in zip(idx, sds):
        x, y, info = o.solve_exact_verified(sd)
        assert info["status"] == 1 and info["polished"] >= 1, (b, info)
        methods[info["method"]] = methods.get(info["method"], 0) + 1
        # KKT conditions of the canonical QP, checked here in numpy: stationarity, primal feasibility, sign of the multipliers, complementarity
        import scipy.sparse as sp
        qp = o.assemble_qp(sd)
        A = sp.csc_matrix((qp["Ax"], qp["Ai"], qp["Ap"]), shape=(o.m, o.n))
        Ax = A @ x; scale = 1.0 + max(np.max(np.abs(Ax)), np.max(np.abs(A.T @ y)))
        assert np.max(np.abs(qp["Pd"] * x + qp["q"] + A.T @ y)) <= 1e-8 * scale
        assert max(np.max(qp["l"] - Ax), np.max(Ax - qp["u"])) <= 1e-8 * scale
        lo = qp["l"] > -1e19; up = qp["u"] < 1e19; eq = qp["l"] == qp["u"]
        assert np.all(y[lo & ~eq] <= 1e-9 * scale) and np.all(y[up & ~eq] >= -1e-9 * scale)
        # complementarity, stated in the two tolerances above: a row is at its bound to the primal tolerance, or its multiplier vanishes to the dual one.  (Round 6: a product
        # bar `|y slack| <= 1e-6 scale` sat here; with the time grid in Julia's range arithmetic -- QP data an ulp away -- two of the ill-conditioned instances are verified by
        # the sparse method with a multiplier of 2e8 on a rate row that is 9e-9 from its bound: product 1.8, both tolerances met.)
        slack = np.where(lo, Ax - qp["l"], qp["u"] - Ax)
        assert np.all((np.abs(slack[~eq]) <= 1e-8 * scale) | (np.abs(y[~eq]) <= 1e-9 * scale)), (b, info)
    assert methods.get("sparse", 0) >= 60 and methods.get("stage+lu", 0) + methods.get("stage", 0) >= 1, methods
