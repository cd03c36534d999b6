"""Partial condensing of the stage-structured LQ problems behind `solve!` (numpy twin; TEST INFRASTRUCTURE, never imported by the product).

SURVEY.md 7.2 / VERDICT r4 item 2: every linear solve of k_solve / k_solve_lat (a Newton step of the interior point, a round of the active-set polish; the QP the
reference hands to OSQP, src/coupled_lat_long.jl:233-309, src/decoupled_lat_long.jl:134-226, solved at src/model_predictive_control.jl:76) is ONE Riccati recursion over
the N stages of the horizon -- a serial chain of N dependent 8 x 8 (5 x 5) matrix steps.  Partial condensing merges m consecutive stages into one:

    x_{k+m} = At x_k + Bt w + ct,      w = (v_k, ..., v_{k+m-1}),      At = A_{k+m-1} ... A_k,   Bt = [A.. B_k | ... | B_{k+m-1}]

with the costs of the skipped nodes expressed in (x_k, w) -- a dense Hessian with cross terms between x_k and w and inside w.  The chain gets m times shorter; a step
gets an (m nv) x (m nv) pivot instead of nv x nv.  This module states both recursions for the general stage form

    min  sum_k 1/2 [x_k; v_k]' [[Q_k, S_k'], [S_k, R_k]] [x_k; v_k] + q_k'x_k + r_k'v_k   +  1/2 x_N'Q_N x_N + q_N'x_N,     x_{k+1} = A_k x_k + B_k v_k + c_k,  x_0 given

so that tests/test_condense_numpy.py can check that they give the same answer on the QP data of BASELINE configs 2 and 5 (empty working set and a penalised one), and
tools/condense_report.py can put numbers on what the condensed pivots look like on the open-loop unstable N = 50 horizon."""
import numpy as np


def riccati(stages, QN, qN, x0):
    """Stage-wise recursion with cross terms.  stages: list of dicts A, B, c, Q, R, S (nv x nx), q, r.  Returns (x [N+1, nx], v list, cond numbers of the pivots)."""
    N = len(stages)
    P, p = QN.copy(), qN.copy()
    K = [None] * N; kf = [None] * N; conds = []
    for k in range(N - 1, -1, -1):
        s = stages[k]
        y = P @ s["c"] + p
        F = s["B"].T @ P @ s["A"] + s["S"]
        Sg = s["R"] + s["B"].T @ P @ s["B"]
        Sg = 0.5 * (Sg + Sg.T)
        conds.append(np.linalg.cond(Sg))
        f = s["r"] + s["B"].T @ y
        L = np.linalg.cholesky(Sg)
        sol = lambda rhs: np.linalg.solve(L.T, np.linalg.solve(L, rhs))
        K[k] = -sol(F); kf[k] = -sol(f)
        Pn = s["Q"] + s["A"].T @ P @ s["A"] + F.T @ K[k]
        p = s["q"] + s["A"].T @ y + F.T @ kf[k]
        P = 0.5 * (Pn + Pn.T)
    x = [np.asarray(x0, float)]; v = []
    for k in range(N):
        s = stages[k]
        vk = K[k] @ x[k] + kf[k]
        v.append(vk); x.append(s["A"] @ x[k] + s["B"] @ vk + s["c"])
    return np.array(x), v, np.array(conds[::-1])


def condense(stages, m):
    """Merge blocks of m consecutive stages (the last block may be shorter).  Returns the condensed stage list and, per block, what is needed to recover the skipped nodes."""
    out, rec = [], []
    N = len(stages)
    for k0 in range(0, N, m):
        blk = stages[k0:k0 + m]
        nx = blk[0]["A"].shape[0]
        nvs = [s["B"].shape[1] for s in blk]; nw = sum(nvs)
        # x_{k0+j} = Phi_j x + Gam_j w + gam_j
        Phi = np.eye(nx); Gam = np.zeros((nx, nw)); gam = np.zeros(nx)
        Q = np.zeros((nx, nx)); S = np.zeros((nw, nx)); R = np.zeros((nw, nw)); q = np.zeros(nx); r = np.zeros(nw)
        maps = []
        off = 0
        for j, s in enumerate(blk):
            maps.append((Phi.copy(), Gam.copy(), gam.copy()))
            E = np.zeros((nvs[j], nw)); E[:, off:off + nvs[j]] = np.eye(nvs[j])         # v_j = E w
            # cost of this stage at (x_j, v_j) with x_j = Phi x + Gam w + gam
            Q += Phi.T @ s["Q"] @ Phi
            S += Gam.T @ s["Q"] @ Phi + E.T @ s["S"] @ Phi
            R += Gam.T @ s["Q"] @ Gam + E.T @ s["R"] @ E + E.T @ s["S"] @ Gam + Gam.T @ s["S"].T @ E
            q += Phi.T @ (s["Q"] @ gam + s["q"])
            r += Gam.T @ (s["Q"] @ gam + s["q"]) + E.T @ (s["S"] @ gam + s["r"])
            # advance
            Phi, Gam, gam = s["A"] @ Phi, s["A"] @ Gam + s["B"] @ E, s["A"] @ gam + s["c"]
            off += nvs[j]
        out.append(dict(A=Phi, B=Gam, c=gam, Q=Q, R=0.5 * (R + R.T), S=S, q=q, r=r))
        rec.append((maps, nvs))
    return out, rec


def riccati_condensed(stages, QN, qN, x0, m):
    """The same problem through blocks of m stages: returns (x [N+1, nx], v list of per-stage inputs, pivot condition numbers [blocks])."""
    cs, rec = condense(stages, m)
    xb, wb, conds = riccati(cs, QN, qN, x0)
    x = []; v = []
    for b, (maps, nvs) in enumerate(rec):
        off = 0
        for j, (Phi, Gam, gam) in enumerate(maps):
            x.append(Phi @ xb[b] + Gam @ wb[b] + gam)
            v.append(wb[b][off:off + nvs[j]]); off += nvs[j]
    x.append(xb[-1])
    return np.array(x), v, conds


# ---- the two stage forms of the library, as general LQ stages (cost of node k+1 moved to where the general form wants it: on x_{k+1} = stage k+1's Q, the last one = Q_N) ----

def coupled_stages(S, cp, rho=0.0, held=None, lam=None):
    """Coupled form (k_solve): x = (q[6], u[2]), v = du.  S: Oracle.unpack_sd dict.  held: optional [N, 16] bool working set (rows 0..5 bounds on x_{k+1}, 12/13 rate rows on v_k;
    soft rows are left out: their slacks are eliminated stage-locally in the kernel and do not change the structure) penalised with rho (augmented Lagrangian, multipliers lam)."""
    N = S["A"].shape[0]
    dt = S["dt"]
    st = []
    Qn = [np.zeros((8, 8)) for _ in range(N + 1)]; qn = [np.zeros(8) for _ in range(N + 1)]
    for k in range(N):
        Ab = np.zeros((8, 8)); Bb = np.zeros((8, 2)); cb = np.zeros(8)
        Ab[:6, :6] = S["A"][k]; Ab[:6, 6:] = S["B0"][k] + S["Bf"][k]; Ab[6:, 6:] = np.eye(2)
        Bb[:6] = S["Bf"][k]; Bb[6:] = np.eye(2); cb[:6] = S["c"][k]
        R = np.diag([2 * cp["R_ddelta"] / dt[k], 2 * cp["R_dFx"] / dt[k]]); r = np.zeros(2)
        Qd = np.zeros(8)
        Qd[0] = 2 * cp["Q_ds"] * dt[k]; Qd[4] = 2 * cp["Q_dpsi"] * dt[k]; Qd[5] = 2 * cp["Q_e"] * dt[k]; Qd[6] = 2 * cp["R_delta"] * dt[k]; Qd[7] = 2 * cp["R_Fx"] * dt[k]
        Qn[k + 1] += np.diag(Qd)
        if held is not None and rho > 0:
            rows = {3: (6, +1.0, S["dmax"][k]), 4: (6, -1.0, -S["dmin"][k]), 5: (7, +1.0, S["fxmax"][k])}
            for j, (idx, sgn, b) in rows.items():
                if held[k, j]:      # t = b - sgn x[idx] held at 0:  rho/2 t^2 - y t
                    Qn[k + 1][idx, idx] += rho; qn[k + 1][idx] += -sgn * rho * b + sgn * (lam[k, j] if lam is not None else 0.0)
            for j, sgn, b in ((12, +1.0, S["ddmax"][k]), (13, -1.0, -S["ddmin"][k])):
                if held[k, j]:
                    R[0, 0] += rho; r[0] += -sgn * rho * b + sgn * (lam[k, j] if lam is not None else 0.0)
        st.append(dict(A=Ab, B=Bb, c=cb, R=R, S=np.zeros((2, 8)), r=r))
    for k in range(N):
        st[k]["Q"] = Qn[k]; st[k]["q"] = qn[k]
    x0 = np.concatenate([S["q_curr"], S["u_curr"]])
    return st, Qn[N], qn[N], x0


def lateral_stages(D, rho=0.0, held=None):
    """Lateral form (k_solve_lat): x = (Uy, r, dpsi, e, delta), v = d-delta.  D: oracle.lat_ipm_numpy.stage_data dict.  held: optional [N, NR] bool (rows 0, 1 on delta_{k+1}, 8, 9 on v_k)."""
    N = D["N"]
    st = []
    Qn = [np.zeros((5, 5)) for _ in range(N + 1)]; qn = [np.zeros(5) for _ in range(N + 1)]
    for k in range(N):
        R = np.array([[D["Rv"][k]]]); r = np.zeros(1)
        Qn[k + 1] += np.diag([0.0, 0.0, D["Qpsi"][k], D["Qe"][k], D["Qd"][k]])
        if held is not None and rho > 0:
            for j, sgn in ((0, +1.0), (1, -1.0)):
                if held[k, j]:
                    Qn[k + 1][4, 4] += rho; qn[k + 1][4] += -sgn * rho * D["b"][k, j]
            for j, sgn in ((8, +1.0), (9, -1.0)):
                if held[k, j]:
                    R[0, 0] += rho; r[0] += -sgn * rho * D["b"][k, j]
        st.append(dict(A=D["Ab"][k], B=D["Bb"][k][:, None], c=D["cb"][k], R=R, S=np.zeros((1, 5)), r=r))
    for k in range(N):
        st[k]["Q"] = Qn[k]; st[k]["q"] = qn[k]
    return st, Qn[N], qn[N], D["x0"]
