#!/usr/bin/env python3
"""Design prototype (numpy, scalar loops) of the structure-exploiting QP solve that the HIP kernel implements.

NOT the oracle and NOT the product: a readable statement of the algorithm in pigeon.jl_amd/csrc/solve_kernel.hip so that
the design can be checked against oracle.solve_exact on the same stage data.  See DESIGN.md section "K8 qp_solve".

Formulation: state x_k = (q_k, u_k) in R^8, input v_k = u_{k+1} - u_k (the reference's d_delta, d_Fx variables,
coupled_lat_long.jl:237-238,244-245), x_{k+1} = Abar_k x_k + Bbar_k v_k + cbar_k, x_0 fixed (:250-251).  Slack variables sigma
(:235-236) are eliminated stage-locally inside every Newton step.  Mehrotra predictor-corrector; each Newton step is ONE
equality-constrained LQ problem in the full point z+ solved by a Riccati recursion (matrix pass once, vector pass twice).
"""
import numpy as np

NROW = 16


def build_rows(S, k, cp, fxmin_n, Ns):
    """Rows a'w <= b for transition k on w = (x_{k+1}[8], v_k[2], s1, s2, sh) -> (Aw [16 x 13], b [16], mask [16])."""
    A = np.zeros((NROW, 13)); b = np.zeros(NROW); mask = np.ones(NROW, bool)
    A[0, 1] = -1; b[0] = -cp["V_min"]
    A[1, 1] = 1; b[1] = cp["V_max"]
    A[2, 7] = -1; b[2] = -fxmin_n
    A[3, 6] = 1; b[3] = S["dmax"][k]
    A[4, 6] = -1; b[4] = -S["dmin"][k]
    A[5, 7] = 1; b[5] = S["fxmax"][k]
    for i in range(4):
        A[6 + i, 2] = S["H"][k, i, 0]; A[6 + i, 3] = S["H"][k, i, 1]; A[6 + i, 10 + i // 2] = -1; b[6 + i] = S["G"][k, i]
    A[10, 10] = -1; A[11, 11] = -1
    A[12, 8] = 1; b[12] = S["ddmax"][k]
    A[13, 8] = -1; b[13] = -S["ddmin"][k]
    node = k + 1
    if node < min(int(cp["N_HJI"]), Ns):
        A[14, 6] = -S["M_hji"][0]; A[14, 7] = -S["M_hji"][1]; A[14, 12] = -1; b[14] = S["b_hji"]
        A[15, 12] = -1
    else:
        mask[14] = mask[15] = False
    return A, b, mask


def solve(S, cp, fxmin_n, Ns, max_iter=40, tol=1e-9, verbose=False, mu0=1.0, tau=1e-3, sig0=0.1):
    N = S["A"].shape[0]
    Ab = np.zeros((N, 8, 8)); Bb = np.zeros((N, 8, 2)); cb = np.zeros((N, 8))
    for k in range(N):
        Ab[k, :6, :6] = S["A"][k]; Ab[k, :6, 6:] = S["B0"][k] + S["Bf"][k]; Ab[k, 6:, 6:] = np.eye(2)
        Bb[k, :6] = S["Bf"][k]; Bb[k, 6:] = np.eye(2); cb[k, :6] = S["c"][k]
    dt = S["dt"]
    Qd = np.zeros((N, 8)); Rd = np.zeros((N, 2)); lin = np.zeros((N, 13))
    for k in range(N):
        Qd[k, 0] = 2 * cp["Q_ds"] * dt[k]; Qd[k, 4] = 2 * cp["Q_dpsi"] * dt[k]; Qd[k, 5] = 2 * cp["Q_e"] * dt[k]
        Qd[k, 6] = 2 * cp["R_delta"] * dt[k]; Qd[k, 7] = 2 * cp["R_Fx"] * dt[k]
        Rd[k] = [2 * cp["R_ddelta"] / dt[k], 2 * cp["R_dFx"] / dt[k]]
        lin[k, 10] = cp["W_beta"] * dt[k]; lin[k, 11] = cp["W_r"] * dt[k]; lin[k, 12] = cp["W_HJI"]
    rows = [build_rows(S, k, cp, fxmin_n, Ns) for k in range(N)]
    x0 = np.concatenate([S["q_curr"], S["u_curr"]])

    # initial point: v = 0 rollout, slacks sigma just feasible, t = max(slack, tau), lambda = mu0 / t
    x = np.zeros((N + 1, 8)); v = np.zeros((N, 2)); sg = np.zeros((N, 3))
    x[0] = x0
    for k in range(N):
        x[k + 1] = Ab[k] @ x[k] + Bb[k] @ v[k] + cb[k]
    t = np.ones((N, NROW)); lam = np.ones((N, NROW))
    for k in range(N):
        A, b, mask = rows[k]
        w = np.concatenate([x[k + 1], v[k], [0, 0, 0]])
        s = b - A @ w
        sg[k, 0] = max(0.0, -min(s[6], s[7])) + sig0
        sg[k, 1] = max(0.0, -min(s[8], s[9])) + sig0
        sg[k, 2] = (max(0.0, -s[14]) + sig0) if mask[14] else 0.0
        w[10:] = sg[k]
        s = b - A @ w
        t[k] = np.maximum(s, tau)
        lam[k] = mu0 / t[k]
        t[k][~mask] = 1.0; lam[k][~mask] = 0.0
    nact = sum(m.sum() for _, _, m in rows)
    phi = 1.0
    rp0 = max(np.abs((rows[k][1] - rows[k][0] @ np.concatenate([x[k + 1], v[k], sg[k]]) - t[k])[rows[k][2]]).max() for k in range(N))

    def newton(ell, Wt):
        """Solve the LQ problem for the full point; returns x+, v+, sg+."""
        Qh = np.zeros((N + 1, 8, 8)); qh = np.zeros((N + 1, 8)); Rh = np.zeros((N, 2, 2)); rh = np.zeros((N, 2))
        elim = []
        for k in range(N):
            A, b, mask = rows[k]
            Phi = A.T @ (Wt[k][:, None] * A)
            Phi[np.arange(8), np.arange(8)] += Qd[k]
            Phi[8, 8] += Rd[k, 0]; Phi[9, 9] += Rd[k, 1]
            g = lin[k] + A.T @ ell[k]
            if not mask[14]:
                Phi[12, 12] = 1.0; g[12] = 0.0
            # eliminate sigma (10, 11, 12): diagonal block
            d = np.array([Phi[10, 10], Phi[11, 11], Phi[12, 12]])
            C = Phi[10:13, :10]                      # 3 x 10
            gs = g[10:13]
            Phi_r = Phi[:10, :10] - C.T @ (C / d[:, None])
            g_r = g[:10] - C.T @ (gs / d)
            elim.append((d, C, gs))
            Qh[k + 1] = Phi_r[:8, :8]; qh[k + 1] = g_r[:8]; Rh[k] = Phi_r[8:, 8:]; rh[k] = g_r[8:]
            assert np.abs(Phi_r[:8, 8:]).max() == 0
        # Riccati
        P = Qh[N].copy(); p = qh[N].copy()
        K = np.zeros((N, 2, 8)); kff = np.zeros((N, 2))
        for k in range(N - 1, -1, -1):
            MA = P @ Ab[k]; MB = P @ Bb[k]
            Sm = Rh[k] + Bb[k].T @ MB
            F = Bb[k].T @ MA
            mc = P @ cb[k] + p
            f = rh[k] + Bb[k].T @ mc
            Si = np.linalg.inv(Sm)
            K[k] = -Si @ F; kff[k] = -Si @ f
            Pn = Qh[k] + Ab[k].T @ MA + F.T @ K[k]
            pn = qh[k] + Ab[k].T @ mc + F.T @ kff[k]
            P = 0.5 * (Pn + Pn.T); p = pn
        xp = np.zeros((N + 1, 8)); vp = np.zeros((N, 2)); sp = np.zeros((N, 3))
        xp[0] = x0
        for k in range(N):
            vp[k] = K[k] @ xp[k] + kff[k]
            xp[k + 1] = Ab[k] @ xp[k] + Bb[k] @ vp[k] + cb[k]
            d, C, gs = elim[k]
            sp[k] = -(C @ np.concatenate([xp[k + 1], vp[k]]) + gs) / d
        return xp, vp, sp

    it = 0
    for it in range(max_iter):
        mu = sum((t[k] * lam[k])[rows[k][2]].sum() for k in range(N)) / nact
        if verbose:
            print(it, "mu", mu, "phi", phi)
        if mu <= tol and phi * max(rp0, 1.0) <= tol:
            break
        Wt = lam / t
        bvec = np.array([rows[k][1] for k in range(N)])
        # full-point form of the linearised complementarity: lambda+ = W C z+ + ell,  ell = (sig mu - corr)/t + lambda - W d
        ell = lam - Wt * bvec                                   # predictor: sig = 0, corr = 0
        xa, va, sa = newton(ell, Wt)
        dta = np.zeros_like(t); dla = np.zeros_like(t)
        for k in range(N):
            A, b, mask = rows[k]
            tp = b - A @ np.concatenate([xa[k + 1], va[k], sa[k]])
            lp = ell[k] + Wt[k] * (b - tp)                      # = W C z+ + ell
            dta[k] = tp - t[k]; dla[k] = lp - lam[k]
            dta[k][~mask] = 0; dla[k][~mask] = 0
        def steplen(dt_, dl_):
            a = 1.0
            for k in range(N):
                m = rows[k][2]
                n1 = m & (dt_[k] < 0); n2 = m & (dl_[k] < 0)
                if n1.any(): a = min(a, (-t[k][n1] / dt_[k][n1]).min())
                if n2.any(): a = min(a, (-lam[k][n2] / dl_[k][n2]).min())
            return a
        aa = steplen(dta, dla)
        mu_aff = sum(((t[k] + aa * dta[k]) * (lam[k] + aa * dla[k]))[rows[k][2]].sum() for k in range(N)) / nact
        sig = (mu_aff / mu) ** 3
        ell = (sig * mu - dta * dla) / t + lam - Wt * bvec
        xc, vc, sc = newton(ell, Wt)
        dtc = np.zeros_like(t); dlc = np.zeros_like(t)
        for k in range(N):
            A, b, mask = rows[k]
            tp = b - A @ np.concatenate([xc[k + 1], vc[k], sc[k]])
            lp = ell[k] + Wt[k] * (b - tp)
            dtc[k] = tp - t[k]; dlc[k] = lp - lam[k]
            dtc[k][~mask] = 0; dlc[k][~mask] = 0
        a = min(1.0, 0.995 * steplen(dtc, dlc))
        x += a * (xc - x); v += a * (vc - v); sg += a * (sc - sg); t += a * dtc; lam += a * dlc
        phi *= (1 - a)
    return dict(x=x, v=v, sg=sg, t=t, lam=lam, iters=it, mu=mu, phi=phi)


if __name__ == "__main__":
    import sys
    sys.path.insert(0, "/root/repo")
    from oracle.oracle import Oracle
    d = np.load("/root/repo/tests/golden/paths/skidpadoval.npz")
    s = d["s_m"]; V = d["UxDes_mps"]
    tt = np.concatenate([[0], np.cumsum(2 * np.diff(s) / (V[:-1] + V[1:]))])
    traj = np.stack([tt, s, V, d["AxDes_mps2"], d["posE_m"], d["posN_m"], d["psi_rad"], d["k_1pm"], d["grade_rad"], 0 * s, d["edgeL_m"], d["edgeR_m"]])
    o = Oracle(); o.set_trajectory(traj)
    cp = o.control_params(); veh = o.vehicle()
    fxmin_n = veh["Fx_min"] / o.u_norm[1]
    rng = np.random.default_rng(12345)
    worst = 0; its = []
    for trial in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
        sq = rng.uniform(5, s[-1] - 60)
        tn = o.traj_at_s(sq); E, Nn, psi = tn[4], tn[5], tn[6]
        e = rng.uniform(-0.5, 0.5)
        Vp = tn[2]
        q0 = np.array([E - e * np.cos(psi), Nn - e * np.sin(psi), psi + rng.uniform(-0.1, 0.1), Vp * rng.uniform(0.9, 1.1), rng.uniform(-0.2, 0.2),
                       tn[7] * Vp + rng.uniform(-0.05, 0.05)])
        Fx0 = rng.uniform(-500, 500)
        u0 = np.array([rng.uniform(-0.05, 0.05), Fx0 * (0.0 if Fx0 > 0 else 0.6), Fx0 * (1.0 if Fx0 > 0 else 0.4)])
        s0, e0, t0, _ = o.path_coordinates(q0[0], q0[1])
        ts, dts = o.time_steps(t0 + rng.uniform(-0.2, 0.2))
        qs, us, ps = o.nodes(q0, u0, ts, dts, time_offset=0.0)
        sd = o.update_qp(qs, us, ps, dts, q0, u0)
        S = o.unpack_sd(sd)
        xe, ye, info = o.solve_exact(sd)
        X = o.split_x(xe)
        R = solve(S, cp, fxmin_n, o.Ns, verbose=(trial == 0))
        du = np.abs(R["x"][:, 6:] - X["u"]).max(); dq = np.abs(R["x"][:, :6] - X["q"]).max()
        dsg = np.abs(R["sg"][:, :2] - X["sigma"]).max()
        worst = max(worst, du); its.append(R["iters"])
        print(trial, "iters", R["iters"], "(exact", info["iters"], ") du", du, "dq", dq, "dsig", dsg, "mu", R["mu"], "phi", R["phi"])
    print("worst du", worst, "iters mean/max", np.mean(its), np.max(its))
