"""TEST INFRASTRUCTURE (part of oracle/: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it).

Stage-structured interior point for the lateral tracking QP of decoupled_lat_long.jl:134-226 in numpy: Mehrotra predictor-corrector in the 5-state stage form, every Newton
step one Riccati recursion.  Two jobs: (1) it is the numpy twin in which the arithmetic of the product's k_solve_lat (pigeon.jl_amd/csrc/pg_solve_lat.hip) was worked out
and its start / stop rules tuned (iteration counts do not depend on the hardware); (2) it is the oracle's SECOND exact method for this formulation: the sparse interior
point of oracle/qp.hpp stalls on the few percent of N = 50 instances whose optimum leaves the linearisation by kilometres (saturated steering on an open-loop unstable
horizon; the regularised KKT factorisation loses the dynamics rows), this recursion -- dynamics exact at every iterate -- does not.  Whatever it returns is only a
CANDIDATE: OracleDecoupled.solve_exact_verified hands it to the canonical QP's active-set polish (oracle/qp.hpp: polish_from), which accepts nothing but a verified KKT
point of the canonical (P, q, A, l, u).  tests/test_lat_ipm_numpy.py checks it against the sparse solver where both work.

Stage form (exact): x_k = (Uy, r, dpsi, e, delta)_k, k = 0..N;  v_k = delta_{k+1} - delta_k (the reference's d-delta variables, :146);
  x_{k+1} = Abar_k x_k + Bbar_k v_k + cbar_k,  Abar = [A  B0+Bf; 0 1],  Bbar = [Bf; 1],  cbar = [c; 0],  x_0 = (q_curr, delta_curr) fixed (:150-151).
Rows of transition k (on node k+1 and v_k), slack t_j >= 0:
  0: dmax - delta     1: delta - dmin     2..5: G_i - H_i (Uy, r) + sigma_{1,1,2,2}     6: sigma_1     7: sigma_2     8: ddmax - v     9: v - ddmin
  walls (build-defined extension): 10: edge_L - e + sw     11: e - edge_R + sw     12: sw
Cost (no 1/2, :213-218): sum_k dt_k (Q_dpsi dpsi^2 + Q_e e^2 + R_delta delta^2)_{k+1} + R_ddelta v_k^2 / dt_k + dt_k (W_beta sigma_1 + W_r sigma_2) [+ wall_weight dt_k sw_k].
Slacks sigma are eliminated stage-locally inside every Newton step; every Newton step is one Riccati recursion (matrix pass once, vector pass for the corrector)."""
import numpy as np


def stage_data(S, cp, walls=None, wall_weight=1000.0):
    """S: OracleDecoupled.unpack_sd dict.  Returns the dict of per-stage arrays the solver works on."""
    N = len(S["dt"])
    Ab = np.zeros((N, 5, 5)); Bb = np.zeros((N, 5)); cb = np.zeros((N, 5))
    Ab[:, :4, :4] = S["A"]; Ab[:, :4, 4] = S["B0"] + S["Bf"]; Ab[:, 4, 4] = 1.0
    Bb[:, :4] = S["Bf"]; Bb[:, 4] = 1.0; cb[:, :4] = S["c"]
    dt = S["dt"]
    NR = 13 if walls is not None else 10
    b = np.zeros((N, NR)); b[:, 0] = S["dmax"]; b[:, 1] = -S["dmin"]; b[:, 2:6] = S["G"]; b[:, 8] = S["ddmax"]; b[:, 9] = -S["ddmin"]
    if walls is not None:
        b[:, 10] = walls[:, 0]; b[:, 11] = -walls[:, 1]
    D = dict(N=N, NR=NR, Ab=Ab, Bb=Bb, cb=cb, b=b, h0=S["H"][:, :, 0].copy(), h1=S["H"][:, :, 1].copy(), dt=dt,
             Qpsi=2 * cp["Q_dpsi"] * dt, Qe=2 * cp["Q_e"] * dt, Qd=2 * cp["R_delta"] * dt, Rv=2 * cp["R_ddelta"] / dt,
             wb=cp["W_beta"] * dt, wr=cp["W_r"] * dt, ww=wall_weight * dt, x0=np.concatenate([S["q_curr"], [S["d_curr"]]]), walls=walls is not None)
    return D


def slacks(D, x, v, s1, s2, sw):
    """t[N, NR] at the point (x[N+1, 5], v[N], sigma)"""
    xn = x[1:]
    t = np.zeros((D["N"], D["NR"]))
    t[:, 0] = D["b"][:, 0] - xn[:, 4]; t[:, 1] = xn[:, 4] + D["b"][:, 1]
    for i in range(4):
        t[:, 2 + i] = D["b"][:, 2 + i] - (D["h0"][:, i] * xn[:, 0] + D["h1"][:, i] * xn[:, 1]) + (s1 if i < 2 else s2)
    t[:, 6] = s1; t[:, 7] = s2; t[:, 8] = D["b"][:, 8] - v; t[:, 9] = v + D["b"][:, 9]
    if D["walls"]:
        t[:, 10] = D["b"][:, 10] - xn[:, 3] + sw; t[:, 11] = D["b"][:, 11] + xn[:, 3] + sw; t[:, 12] = sw
    return t


def assemble(D, W, ell):
    """Barrier (or penalty) terms of every stage -> (Qhat[N,5,5], qhat[N,5], Rhat[N], rhat[N], elimination coefficients)."""
    N = D["N"]; h0, h1 = D["h0"], D["h1"]
    Qh = np.zeros((N, 5, 5)); qh = np.zeros((N, 5))
    E = {}
    # slack group 1 (rows 2, 3, pivot 6), group 2 (rows 4, 5, pivot 7)
    for g, (r0, r1, rp, w) in enumerate([(2, 3, 6, D["wb"]), (4, 5, 7, D["wr"])]):
        d = 1.0 / (W[:, r0] + W[:, r1] + W[:, rp])
        c0 = -(W[:, r0] * h0[:, r0 - 2] + W[:, r1] * h0[:, r1 - 2]); c1 = -(W[:, r0] * h1[:, r0 - 2] + W[:, r1] * h1[:, r1 - 2])
        gg = w - ell[:, r0] - ell[:, r1] - ell[:, rp]
        E[g] = (c0, c1, gg, d)
        Qh[:, 0, 0] -= c0 * c0 * d; Qh[:, 0, 1] -= c0 * c1 * d; Qh[:, 1, 1] -= c1 * c1 * d
        qh[:, 0] -= c0 * gg * d; qh[:, 1] -= c1 * gg * d
    for i in range(4):
        Qh[:, 0, 0] += W[:, 2 + i] * h0[:, i] ** 2; Qh[:, 0, 1] += W[:, 2 + i] * h0[:, i] * h1[:, i]; Qh[:, 1, 1] += W[:, 2 + i] * h1[:, i] ** 2
        qh[:, 0] += h0[:, i] * ell[:, 2 + i]; qh[:, 1] += h1[:, i] * ell[:, 2 + i]
    Qh[:, 1, 0] = Qh[:, 0, 1]
    Qh[:, 2, 2] = D["Qpsi"]; Qh[:, 3, 3] = D["Qe"]; Qh[:, 4, 4] = D["Qd"] + W[:, 0] + W[:, 1]
    qh[:, 4] = ell[:, 0] - ell[:, 1]
    if D["walls"]:
        d = 1.0 / (W[:, 10] + W[:, 11] + W[:, 12]); ce = -(W[:, 10] - W[:, 11]); gg = D["ww"] - ell[:, 10] - ell[:, 11] - ell[:, 12]
        E[2] = (ce, gg, d)
        Qh[:, 3, 3] += W[:, 10] + W[:, 11] - ce * ce * d
        qh[:, 3] = (ell[:, 10] - ell[:, 11]) - ce * gg * d
    Rh = D["Rv"] + W[:, 8] + W[:, 9]; rh = ell[:, 8] - ell[:, 9]
    return Qh, qh, Rh, rh, E


def riccati_matrices(D, Qh, qh, Rh, rh):
    """Backward matrix pass with the predictor's vector recursion riding along.  Stage cost k sits on node k+1."""
    N = D["N"]
    K = np.zeros((N, 5)); Si = np.zeros(N); Mc = np.zeros((N, 5)); kff = np.zeros(N)
    P = Qh[N - 1].copy(); p = qh[N - 1].copy()
    for k in range(N - 1, -1, -1):
        Ps = 0.5 * (P + P.T)                                  # the kernel uses P from both orientations: the antisymmetric rounding error never propagates
        A, B, c = D["Ab"][k], D["Bb"][k], D["cb"][k]
        MA = Ps @ A; MB = Ps @ B; Mc[k] = Ps @ c
        y = Mc[k] + p
        F = B @ MA; S = Rh[k] + B @ MB; f = rh[k] + B @ y
        Si[k] = 1.0 / S
        K[k] = -F * Si[k]; kff[k] = -f * Si[k]
        if k > 0:
            P = Qh[k - 1] + A.T @ MA + np.outer(F, K[k])
            p = qh[k - 1] + A.T @ y + F * kff[k]
    return K, Si, Mc, kff


def riccati_vectors(D, qh, rh, K, Si, Mc):
    N = D["N"]
    kff = np.zeros(N); p = qh[N - 1].copy()
    for k in range(N - 1, -1, -1):
        A, B = D["Ab"][k], D["Bb"][k]
        y = Mc[k] + p
        f = rh[k] + B @ y
        kff[k] = -Si[k] * f
        if k > 0:
            p = qh[k - 1] + A.T @ y + K[k] * f
    return kff


def forward(D, K, kff, use_gain=True):
    N = D["N"]
    x = np.zeros((N + 1, 5)); v = np.zeros(N); x[0] = D["x0"]
    for k in range(N):
        v[k] = K[k] @ x[k] + kff[k] if use_gain else 0.0
        x[k + 1] = D["Ab"][k] @ x[k] + D["Bb"][k] * v[k] + D["cb"][k]
    return x, v


def eliminated_slacks(D, E, x):
    xn = x[1:]
    c0, c1, gg, d = E[0]; s1 = -(c0 * xn[:, 0] + c1 * xn[:, 1] + gg) * d
    c0, c1, gg, d = E[1]; s2 = -(c0 * xn[:, 0] + c1 * xn[:, 1] + gg) * d
    sw = np.zeros(D["N"])
    if D["walls"]:
        ce, gg, d = E[2]; sw = -(ce * xn[:, 3] + gg) * d
    return s1, s2, sw


def solve(D, tol=1e-12, mu0=100.0, max_iter=40, sig0=1.0, tau=1e-4, floor=True, verbose=False, start="rollout", wls=1.0, mu0_cost=10.0, split_steps=False):
    """Returns dict(x, v, s1, s2, sw, t, lam, iters, status, mu).  status 1 solved, 2 iteration cap, 4 numerical.
    mu0_cost: the first barrier parameter is max(mu0, mu0_cost x cost of the starting point per row) as in k_solve_lat (0: mu0 as given).
    start: "rollout" = v = 0 roll-out (what the kernel does); "ls" = one Newton solve with every row replaced by a quadratic penalty of weight wls (closed-loop roll-out,
    bounded on the open-loop unstable horizons), then a uniform shift that makes every slack >= 1."""
    N, NR = D["N"], D["NR"]
    z = np.zeros(N)
    if start == "rollout":
        x, v = forward(D, None, None, use_gain=False)                      # v = 0 roll-out
        sl = slacks(D, x, v, z, z, z)
        s1 = np.maximum(0.0, -np.minimum(sl[:, 2], sl[:, 3])) + sig0; s2 = np.maximum(0.0, -np.minimum(sl[:, 4], sl[:, 5])) + sig0
        sw = np.maximum(0.0, -np.minimum(sl[:, 10], sl[:, 11])) + sig0 if D["walls"] else z
        sl = slacks(D, x, v, s1, s2, sw)
        xn = x[1:]
        j0 = 0.5 * float(np.sum(D["Qpsi"] * xn[:, 2] ** 2 + D["Qe"] * xn[:, 3] ** 2 + D["Qd"] * xn[:, 4] ** 2)) + float(np.sum(D["wb"] * s1 + D["wr"] * s2 + (D["ww"] * sw if D["walls"] else 0.0)))
        mu0 = max(mu0, mu0_cost * j0 / (N * NR))
        t = np.maximum(sl, tau); lam = mu0 / t
        rp0 = float(np.max(t - sl))
    else:
        W = np.full((N, NR), wls); ell = W * 0.0 + wls * 1.0 - W * D["b"]       # lambda = t = 1: ell = lambda - W b
        Qh, qh, Rh, rh, E = assemble(D, W, ell)
        K, Si, Mc, kff = riccati_matrices(D, Qh, qh, Rh, rh)
        x, v = forward(D, K, kff)
        s1, s2, sw = eliminated_slacks(D, E, x)
        sl = slacks(D, x, v, s1, s2, sw)
        shift = max(0.0, 1.0 - float(sl.min()))
        t = sl + shift; lam = mu0 / t
        rp0 = shift
    phi = 1.0
    ntot = N * NR
    status = 2; it = 0; good = 0; mu = 0.0
    while True:
        if it >= max_iter and not (max_iter >= 20 and good >= 3 and it < max_iter + 20):
            break
        mu = float(np.sum(t * lam)) / ntot
        if not np.isfinite(mu):
            status = 4; break
        if mu <= tol and phi * max(rp0, 1.0) <= tol:
            status = 1; break
        it_ = 1.0 / t; W = lam * it_
        # predictor
        ell = lam - W * D["b"]
        Qh, qh, Rh, rh, E = assemble(D, W, ell)
        K, Si, Mc, kff = riccati_matrices(D, Qh, qh, Rh, rh)
        xn, vn = forward(D, K, kff)
        n1, n2, nw = eliminated_slacks(D, E, xn)
        tp = slacks(D, xn, vn, n1, n2, nw)
        dt_ = tp - t; dl_ = -W * tp
        corr = dt_ * dl_
        rmax = float(np.max(np.maximum(-dt_ * it_, tp * it_)))          # -dl/lam = tp/t for the affine direction
        aaff = 1.0 / rmax if rmax > 1.0 else 1.0
        if floor and mu <= 1e4 * tol and aaff < 0.3 and phi * max(rp0, 1.0) <= tol:
            status = 1; break
        mu_aff = float(np.sum((t + aaff * dt_) * (lam + aaff * dl_))) / ntot
        sg = min(mu_aff / mu, 1.0) ** 3
        # corrector
        ell = (sg * mu - corr) * it_ + lam - W * D["b"]
        _, qh, _, rh, E = assemble(D, W, ell)
        kff = riccati_vectors(D, qh, rh, K, Si, Mc)
        xn, vn = forward(D, K, kff)
        n1, n2, nw = eliminated_slacks(D, E, xn)
        tp = slacks(D, xn, vn, n1, n2, nw)
        dt_ = tp - t; dl_ = (sg * mu - corr) * it_ - W * tp
        rmax = float(np.max(np.maximum(-dt_ * it_, -dl_ / lam)))
        alpha = 0.995 / rmax if rmax > 0.995 else 1.0
        alpha_d = alpha
        if split_steps:          # (experiment of round 4: the slacks and the multipliers each go to 0.995 of their OWN boundary)
            rp = float(np.max(-dt_ * it_)); rd = float(np.max(-dl_ / lam))
            alpha = 0.995 / rp if rp > 0.995 else 1.0; alpha_d = 0.995 / rd if rd > 0.995 else 1.0
        if floor and mu <= 1e5 * tol and phi * max(rp0, 1.0) <= tol:
            mnew = float(np.sum((t + alpha * dt_) * (lam + alpha_d * dl_))) / ntot
            if not (mnew <= 4.0 * mu):
                status = 1; break
        t = t + alpha * dt_; lam = lam + alpha_d * dl_
        x = x + alpha * (xn - x); v = v + alpha * (vn - v); s1 = s1 + alpha * (n1 - s1); s2 = s2 + alpha * (n2 - s2); sw = sw + alpha * (nw - sw)
        phi *= (1.0 - alpha)
        good = good + 1 if alpha > 0.5 else 0
        if verbose:
            print(f"it {it:3d} mu {mu:.3e} aaff {aaff:.3f} sg {sg:.2e} alpha {alpha:.3f} alpha_d {alpha_d:.3f} phi {phi:.1e}")
        if mu > 1e8 * mu0:
            break
        it += 1
    return dict(x=x, v=v, s1=s1, s2=s2, sw=sw, t=t, lam=lam, iters=it, status=status, mu=mu)


def canonical_candidate(D, r, Ns, walls=False):
    """(x0, actv) for polish_from: the solver's answer `r` of stage data D in the variable and row order of the canonical lateral QP (oracle/mpc_decoupled.hpp =
    the @constraint order of decoupled_lat_long.jl:146-203; with walls=True the N slack columns and 3N rows oracle.extend_with_walls appends)."""
    N = D["N"]; Nn = N + 1
    x0 = np.concatenate([r["x"][:, :4].ravel(), r["x"][:, 4], np.stack([r["s1"], r["s2"]], axis=1).ravel(), r["v"]] + ([r["sw"]] if walls else []))
    m = 15 * N + 5
    actv = np.zeros(m + (3 * N if walls else 0), dtype=np.int32)
    on = r["lam"] > r["t"]
    base = 2 * N + N + 5 + 4 * N
    for k in range(N):
        actv[2 * k] = on[k, 6]; actv[2 * k + 1] = on[k, 7]                     # vec(sigma) >= 0, column-major (:147)
        b = base + 8 * k                                                        # per transition (:190-194): delta <= max, delta >= min, 4 envelope rows, d-delta <= max, >= min
        actv[b] = on[k, 0]; actv[b + 1] = on[k, 1]; actv[b + 2:b + 6] = on[k, 2:6]; actv[b + 6] = on[k, 8]; actv[b + 7] = on[k, 9]
        if walls:
            actv[m + 3 * k] = on[k, 10]; actv[m + 3 * k + 1] = on[k, 11]; actv[m + 3 * k + 2] = on[k, 12]
    return x0, actv
