"""GPU parity of the decoupled (lateral) formulation (decoupled_lat_long.jl) against the CPU oracle, N = 30 and N = 50 (BASELINE config 5)."""
import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu


def embed_sd(orc, sd, ux_dummy=8.0):
    """Oracle stage data of the lateral QP -> the embedded coupled layout pg_get_qp returns for PG_DECOUPLED handles."""
    S = orc.unpack_sd(sd); N = orc.N
    A = np.zeros((N, 6, 6)); A[:, 0, 0] = 1; A[:, 1, 1] = 1; A[:, 2:, 2:] = S["A"]
    B0 = np.zeros((N, 6, 2)); B0[:, 2:, 0] = S["B0"]; Bf = np.zeros((N, 6, 2)); Bf[:, 2:, 0] = S["Bf"]
    c = np.zeros((N, 6)); c[:, 2:] = S["c"]
    return np.concatenate([A.ravel(), B0.ravel(), Bf.ravel(), c.ravel(), S["H"].ravel(), S["G"].ravel(), S["dmin"], S["dmax"], np.ones(N), S["ddmin"], S["ddmax"],
                           S["dt"], [0.0, ux_dummy], S["q_curr"], [S["d_curr"], 0.0], [0.0, 0.0], [1.0]])


@pytest.mark.parametrize("Ns,Nl", [(10, 20), (10, 40)])
def test_decoupled_matches_oracle(pkg, oracle_mod, skidpad, Ns, Nl):
    B = 64
    mpc = pkg.DecoupledTrajectoryTrackingMPC(pkg.X1(), skidpad, B, N_short=Ns, N_long=Nl)
    assert np.array_equal(mpc.u_normalization, [1.0, 1.0])
    orc = oracle_mod.OracleDecoupled(N_short=Ns, N_long=Nl); orc.set_trajectory(skidpad.data)
    assert (orc.n, orc.m) == ((245, 455) if Nl == 20 else (405, 755))
    state, control, t0, toff = pkg.synthetic.config2_inputs(skidpad, B, seed=31, traj_mode=(Nl == 20))
    u, status, iters = mpc.step_(state, control, t0, time_offset=toff)
    assert np.all(status == 1), status
    qs, us, ps = mpc.nodes(); qp = mpc.qp_data(); x, sg = mpc.solution(); st, it, act, mu = mpc.solve_info()
    worst = 0.0
    for b in range(B):
        ts, dt = orc.time_steps(t0[b])
        oq, ou, op = orc.nodes(state[b], control[b], ts, dt, time_offset=toff[b])
        assert np.max(np.abs(qs[b, :, 2:] - oq)) < 1e-9 and np.max(np.abs(us[b] - ou)) <= 1e-9 * max(1, np.max(np.abs(ou)))
        assert np.max(np.abs(qs[b, :, 1] - op[:, 0])) < 1e-9 and np.max(np.abs(ps[b, :, 1] - op[:, 1])) < 1e-12      # Ux parameter, kappa
        sd = orc.update_qp(oq, ou, op, dt)
        ref = embed_sd(orc, sd)
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
        assert pkg.decoupled_canonical_active_set(orc.N, Ns, act[b]) == oracle_mod.active_set(qpc, xe, ye, tol=1e-6), b
    assert worst < 1e-6, worst


def test_decoupled_refuses_hji(pkg, skidpad):
    mpc = pkg.DecoupledTrajectoryTrackingMPC(pkg.X1(), skidpad, 4)
    knots, V, g = pkg.synthetic.hji_grid(dims=(3, 3, 3, 3, 3, 3, 3))
    with pytest.raises(pkg.PigeonError):
        mpc.set_hji_cache(knots, V, g)
