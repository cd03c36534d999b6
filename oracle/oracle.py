"""ORACLE — TEST INFRASTRUCTURE ONLY.  ctypes front-end of oracle/liboracle.so.

CPU restatement of the reference hot path (see the headers under oracle/ for file:line citations).
PARITY UNPINNED: /root/reference holds no golden vectors for this path (test/runtests.jl:1-5) and cannot
run here (no Julia toolchain, third-party packages absent).  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module; the product path never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import scipy.sparse as sp

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int)
c_fp = C.POINTER(C.c_float)


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".cpp", ".hpp"))]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        _LIB = C.CDLL(so)
        _LIB.po_create.restype = C.c_void_p
        _LIB.po_create.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int]
        _LIB.po_step_batch.restype = C.c_double
    return _LIB


def _d(a):
    return a.ctypes.data_as(c_dp)


def _arr(x, n=None):
    a = np.ascontiguousarray(x, dtype=np.float64)
    if n is not None:
        assert a.size == n, (a.size, n)
    return a


class Oracle:
    """One coupled MPC problem family (N_short, N_long, dt's) on one trajectory."""

    def __init__(self, N_short=10, N_long=20, dt_short=0.01, dt_long=0.2, use_correction_step=True, rk4_substeps=10):
        self.L = lib()
        self.h = C.c_void_p(self.L.po_create(N_short, N_long, dt_short, dt_long, int(use_correction_step), rk4_substeps))
        self.Ns, self.Nl = N_short, N_long
        self.N = N_short + N_long
        self.Nn = self.N + 1
        veh = np.zeros(22); cp = np.zeros(16); un = np.zeros(2)
        self.L.po_get_params(self.h, _d(veh), _d(cp), _d(un))
        self.veh, self.cp, self.u_norm = veh, cp, un
        n = C.c_int(); m = C.c_int(); nnz = C.c_int()
        self.L.po_qp_dims(self.h, C.byref(n), C.byref(m), C.byref(nnz))
        self.n, self.m, self.nnz = n.value, m.value, nnz.value
        self.sd_len = self.L.po_sd_len(self.h)

    def __del__(self):
        try:
            self.L.po_destroy(self.h)
        except Exception:
            pass

    VEH_FIELDS = ["G", "m", "Izz", "L", "a", "b", "h", "mu", "Caf", "Car", "Cd0", "Cd1", "Cd2", "fwd_frac", "rwd_frac", "fwb_frac",
                  "rwb_frac", "Fx_max", "Fx_min", "Px_max", "delta_max", "kappa_max"]
    CP_FIELDS = ["V_min", "V_max", "k_V", "k_s", "deltadot_max", "Q_ds", "Q_dpsi", "Q_e", "W_beta", "W_r", "W_HJI", "N_HJI", "R_delta",
                 "R_ddelta", "R_Fx", "R_dFx"]

    def vehicle(self):
        return dict(zip(self.VEH_FIELDS, self.veh))

    def control_params(self):
        return dict(zip(self.CP_FIELDS, self.cp))

    def set_control_params(self, **kw):
        for k, v in kw.items():
            self.cp[self.CP_FIELDS.index(k)] = v
        self.L.po_set_control_params(self.h, _d(self.cp))

    def set_hji_eps(self, eps):
        self.L.po_set_hji_eps(self.h, C.c_double(eps))

    # ---- data ----
    def set_trajectory(self, traj12):
        a = _arr(traj12)
        assert a.ndim == 2 and a.shape[0] == 12
        self.traj = a
        self.L.po_set_trajectory(self.h, a.shape[1], _d(a))

    def set_hji_grid(self, knots, V, gradV):
        dims = np.array([len(k) for k in knots], dtype=np.int32)
        kc = np.ascontiguousarray(np.concatenate(knots), dtype=np.float32)
        Vf = np.ascontiguousarray(V, dtype=np.float32); gf = np.ascontiguousarray(gradV, dtype=np.float32)
        assert Vf.size == int(np.prod(dims)) and gf.size == 7 * Vf.size
        self.L.po_set_hji_grid(self.h, dims.ctypes.data_as(c_ip), kc.ctypes.data_as(c_fp), Vf.ctypes.data_as(c_fp), gf.ctypes.data_as(c_fp))

    # ---- pieces ----
    def time_steps(self, t0):
        ts = np.zeros(self.Nn); dt = np.zeros(self.N)
        self.L.po_time_steps(self.h, C.c_double(t0), _d(ts), _d(dt))
        return ts, dt

    def set_time_grid_naive(self, naive):
        """True: `t0 + dt*i` with two roundings (rounds 1-5); False (default): Julia's range arithmetic (oracle/julia_range.hpp; model_predictive_control.jl:25-26)."""
        self.L.po_set_time_grid_naive(self.h, int(bool(naive)))

    def simulate_times(self, dt, t_end, steps, t_start=0.0):
        """(t_start .+ (0:dt:t_end))[1:steps] -- the loop variable of `simulate` (model_predictive_control.jl:87) as Julia's range gives it."""
        out = np.zeros(steps); self.L.po_simulate_times.restype = C.c_int
        self.L.po_simulate_times(C.c_double(dt), C.c_double(t_end), C.c_double(t_start), int(steps), _d(out))
        return out

    def path_coordinates(self, E, N):
        out = np.zeros(3); im = C.c_int()
        self.L.po_path_coordinates(self.h, C.c_double(E), C.c_double(N), _d(out), C.byref(im))
        return out[0], out[1], out[2], im.value

    def traj_at_time(self, t):
        o = np.zeros(12); self.L.po_traj_at_time(self.h, C.c_double(t), _d(o)); return o

    def traj_at_s(self, s):
        o = np.zeros(12); self.L.po_traj_at_s(self.h, C.c_double(s), _d(o)); return o

    def tracking_dynamics(self, q, u, p):
        o = np.zeros(6); self.L.po_tracking_dynamics(self.h, _d(_arr(q, 6)), _d(_arr(u, 2)), _d(_arr(p, 4)), _d(o)); return o

    def world_dynamics(self, q, u):
        o = np.zeros(6); self.L.po_world_dynamics(self.h, _d(_arr(q, 6)), _d(_arr(u, 2)), _d(o)); return o

    def stable_limits(self, Ux, Fxf, Fxr):
        o = np.zeros(14); self.L.po_stable_limits(self.h, C.c_double(Ux), C.c_double(Fxf), C.c_double(Fxr), _d(o))
        return o[0], o[1], o[2:10].reshape(4, 2), o[10:14]

    def steady_state(self, V, A_tan, kappa, num_iters=4, r=None, beta0=0.0, delta0=0.0, Fyf0=0.0):
        o = np.zeros(8)
        r = V * kappa if r is None else r
        self.L.po_steady_state(self.h, C.c_double(V), C.c_double(A_tan), C.c_double(kappa), num_iters, C.c_double(r), C.c_double(beta0),
                               C.c_double(delta0), C.c_double(Fyf0), _d(o))
        return dict(zip(["beta", "Ux", "Uy", "r", "A", "delta", "Fxf", "Fxr"], o))

    def linearize_interval(self, q, u0, p0, uf, pf, dt, ramp):
        A = np.zeros((6, 6)); B0 = np.zeros((6, 2)); Bf = np.zeros((6, 2)); c = np.zeros(6)
        self.L.po_linearize_interval(self.h, _d(_arr(q, 6)), _d(_arr(u0, 2)), _d(_arr(p0, 4)), _d(_arr(uf, 2)), _d(_arr(pf, 4)), C.c_double(dt),
                                     int(ramp), _d(A), _d(B0), _d(Bf), _d(c))
        return A, B0, Bf, c

    def propagate_tracking(self, q, u0, p0, uf, pf, dt, ramp):
        x = _arr(q, 6).copy()
        self.L.po_propagate_tracking(self.h, _d(x), _d(_arr(u0, 2)), _d(_arr(p0, 4)), _d(_arr(uf, 2)), _d(_arr(pf, 4)), C.c_double(dt), int(ramp))
        return x

    def plant_step(self, q6, u3, dt):
        x = _arr(q6, 6).copy(); self.L.po_plant_step(self.h, _d(x), _d(_arr(u3, 3)), C.c_double(dt)); return x

    def next_control(self, u2n):
        o = np.zeros(3); self.L.po_next_control(self.h, _d(_arr(u2n, 2)), _d(o)); return o

    def hji_relative_state(self, us6, them4):
        o = np.zeros(7); self.L.po_hji_relative_state(_d(_arr(us6, 6)), _d(_arr(them4, 4)), _d(o)); return o

    def hji_lookup(self, x7):
        V = C.c_double(); g = np.zeros(7)
        inb = self.L.po_hji_lookup(self.h, _d(_arr(x7, 7)), C.byref(V), _d(g))
        return V.value, g, bool(inb)

    def hji_slice(self, knots, q7):
        """rviz.jl:23-40,60-69 restated: V at every knot pair (x, y) of grid dimensions 1, 2 with the other five components of the relative state q7;
        marker colours (value_to_RGB :41-44); zero-level crossings on the grid edges = vertex set of Contour.jl's contour(X, Y, V, 0)
        (an edge carries a vertex iff exactly one end is above the level; linear interpolation along the edge)."""
        X = np.asarray(knots[0], dtype=np.float64); Y = np.asarray(knots[1], dtype=np.float64)
        V = np.zeros((len(X), len(Y)))
        for i, x in enumerate(X):
            for j, y in enumerate(Y):
                V[i, j] = self.hji_lookup([x, y] + [float(v) for v in q7[2:]])[0]
        xx = np.clip(np.where(V < 0, 0.5 * (-3.0 - V) / -3.0, 0.5 + 0.5 * V / 20.0), 0.0, 1.0)
        rgb = (1 - xx)[..., None] * np.array([1.0, 0.5, 0.0]) + xx[..., None] * np.array([0.0, 0.5, 1.0])
        up = V > 0
        with np.errstate(invalid="ignore", divide="ignore"):
            cx = np.where(up[:-1] != up[1:], X[:-1, None] + (0 - V[:-1]) / (V[1:] - V[:-1]) * (X[1:, None] - X[:-1, None]), np.nan)
            cy = np.where(up[:, :-1] != up[:, 1:], Y[None, :-1] + (0 - V[:, :-1]) / (V[:, 1:] - V[:, :-1]) * (Y[None, 1:] - Y[None, :-1]), np.nan)
        return V, rgb, cx, cy

    def hji_optimal_control(self, state6, other4):
        """(V, (delta_opt, Fx_opt)): optimal_control of HJI_computation.jl:133-158 at cache[relative_state].gradV"""
        u2 = np.zeros(2)
        self.L.po_hji_optimal_control.restype = C.c_double
        V = self.L.po_hji_optimal_control(self.h, _d(_arr(state6, 6)), _d(_arr(other4, 4)), _d(u2))
        return float(V), u2

    def hji_constraint(self, state6, other4, control3):
        M = np.zeros(2); b = C.c_double(); V = C.c_double()
        self.L.po_hji_constraint(self.h, _d(_arr(state6, 6)), _d(_arr(other4, 4)), _d(_arr(control3, 3)), _d(M), C.byref(b), C.byref(V))
        return M, b.value, V.value

    def nodes(self, state6, control3, ts, dt, time_offset=float("nan"), solved=False, prev_ts=None, prev_q=None, prev_u=None):
        qs = np.zeros((self.Nn, 6)); us = np.zeros((self.Nn, 2)); ps = np.zeros((self.Nn, 4))
        pt = _arr(prev_ts, self.Nn) if prev_ts is not None else None
        pq = _arr(prev_q, 6 * self.Nn) if prev_q is not None else None
        pu = _arr(prev_u, 2 * self.Nn) if prev_u is not None else None
        self.L.po_nodes(self.h, _d(_arr(state6, 6)), _d(_arr(control3, 3)), C.c_double(time_offset), int(solved), _d(_arr(ts, self.Nn)),
                        _d(_arr(dt, self.N)), _d(pt) if pt is not None else None, _d(pq) if pq is not None else None,
                        _d(pu) if pu is not None else None, _d(qs), _d(us), _d(ps))
        return qs, us, ps

    def update_qp(self, qs, us, ps, dt, state6, control3, other4=(0, 0, 0, 0)):
        sd = np.zeros(self.sd_len); V = C.c_double()
        self.L.po_update_qp(self.h, _d(_arr(qs, 6 * self.Nn)), _d(_arr(us, 2 * self.Nn)), _d(_arr(ps, 4 * self.Nn)), _d(_arr(dt, self.N)),
                            _d(_arr(state6, 6)), _d(_arr(control3, 3)), _d(_arr(other4, 4)), _d(sd), C.byref(V))
        return sd

    def unpack_sd(self, sd):
        N = self.N; o = 0; out = {}
        for name, sz, shp in [("A", 36, (N, 6, 6)), ("B0", 12, (N, 6, 2)), ("Bf", 12, (N, 6, 2)), ("c", 6, (N, 6)), ("H", 8, (N, 4, 2)), ("G", 4, (N, 4)),
                              ("dmin", 1, (N,)), ("dmax", 1, (N,)), ("fxmax", 1, (N,)), ("ddmin", 1, (N,)), ("ddmax", 1, (N,)), ("dt", 1, (N,))]:
            out[name] = sd[o:o + sz * N].reshape(shp); o += sz * N
        out["q_curr"] = sd[o:o + 6]; o += 6; out["u_curr"] = sd[o:o + 2]; o += 2; out["M_hji"] = sd[o:o + 2]; o += 2; out["b_hji"] = sd[o]
        return out

    def pack_sd(self, d):
        return np.concatenate([np.ravel(d[k]) for k in ["A", "B0", "Bf", "c", "H", "G", "dmin", "dmax", "fxmax", "ddmin", "ddmax", "dt", "q_curr", "u_curr",
                                                        "M_hji"]] + [np.atleast_1d(d["b_hji"])]).astype(np.float64)

    def assemble_qp(self, sd):
        Pd = np.zeros(self.n); q = np.zeros(self.n); Ap = np.zeros(self.n + 1, dtype=np.int32); Ai = np.zeros(self.nnz, dtype=np.int32)
        Ax = np.zeros(self.nnz); l = np.zeros(self.m); u = np.zeros(self.m)
        self.L.po_assemble_qp(self.h, _d(_arr(sd, self.sd_len)), _d(Pd), _d(q), Ap.ctypes.data_as(c_ip), Ai.ctypes.data_as(c_ip), _d(Ax), _d(l), _d(u))
        return dict(Pd=Pd, q=q, Ap=Ap, Ai=Ai, Ax=Ax, l=l, u=u)

    def solve_exact(self, sd):
        x = np.zeros(self.n); y = np.zeros(self.m); info = np.zeros(6)
        st = self.L.po_solve_exact(self.h, _d(_arr(sd, self.sd_len)), _d(x), _d(y), _d(info))
        return x, y, dict(iters=int(info[0]), status=int(info[1]), res_pri=info[2], res_dua=info[3], gap=info[4], polished=int(info[5]))

    def osqp_settings(self, rho=0.1, sigma=1e-6, alpha=1.6, eps_abs=1e-3, eps_rel=1e-3, max_iter=4000, scaling=10, check_termination=25,
                      adaptive_rho=1, adaptive_rho_interval=25, warm_start=1):
        self.L.po_osqp_settings(self.h, C.c_double(rho), C.c_double(sigma), C.c_double(alpha), C.c_double(eps_abs), C.c_double(eps_rel), max_iter,
                                scaling, check_termination, adaptive_rho, adaptive_rho_interval, warm_start)

    def osqp_solve(self, sd, inst=0):
        x = np.zeros(self.n); y = np.zeros(self.m); info = np.zeros(6)
        st = self.L.po_osqp_solve(self.h, inst, _d(_arr(sd, self.sd_len)), _d(x), _d(y), _d(info))
        return x, y, dict(iters=int(info[0]), status=int(info[1]), res_pri=info[2], res_dua=info[3], rho=info[4], n_refactor=int(info[5]))

    def reset_instance(self, inst):
        self.L.po_reset_instance(self.h, inst)

    def split_x(self, x):
        Nn, N, Ns = self.Nn, self.N, self.Ns
        o = 0; q = x[o:o + 6 * Nn].reshape(Nn, 6); o += 6 * Nn
        u = x[o:o + 2 * Nn].reshape(Nn, 2); o += 2 * Nn
        sg = x[o:o + 2 * N].reshape(N, 2); o += 2 * N
        sh = x[o:o + Ns]; o += Ns
        dd = x[o:o + N]; o += N
        df = x[o:o + N]
        return dict(q=q, u=u, sigma=sg, sigma_hji=sh, ddelta=dd, dFx=df)

    def step_batch(self, states6, controls3, t0, others4=None, time_offsets=None, solver=0, nthreads=1, want_sol=False):
        B = len(t0)
        s = _arr(states6, 6 * B); c = _arr(controls3, 3 * B); t = _arr(t0, B)
        o = _arr(others4, 4 * B) if others4 is not None else None
        to = _arr(time_offsets, B) if time_offsets is not None else None
        u = np.zeros((B, 3)); sol = np.zeros((B, 8 * self.Nn)) if want_sol else None
        it = np.zeros(B, dtype=np.int32); st = np.zeros(B, dtype=np.int32)
        secs = self.L.po_step_batch(self.h, B, _d(s), _d(c), _d(t), _d(o) if o is not None else None, _d(to) if to is not None else None, solver, nthreads,
                                    _d(u), _d(sol) if sol is not None else None, it.ctypes.data_as(c_ip), st.ctypes.data_as(c_ip))
        return u, sol, it, st, secs


def solve_exact_generic(qp):
    """Exact optimum of a canonical QP dict (Pd, q, Ap, Ai, Ax, l, u) with the oracle's sparse interior point: (x, y, info)."""
    L = lib()
    n = len(qp["Pd"]); m = len(qp["l"])
    x = np.zeros(n); y = np.zeros(m); info = np.zeros(6)
    Ap = np.ascontiguousarray(qp["Ap"], dtype=np.int32); Ai = np.ascontiguousarray(qp["Ai"], dtype=np.int32)
    L.po_solve_exact_generic(n, m, _d(_arr(qp["Pd"])), _d(_arr(qp["q"])), Ap.ctypes.data_as(c_ip), Ai.ctypes.data_as(c_ip), _d(_arr(qp["Ax"])),
                             _d(_arr(qp["l"])), _d(_arr(qp["u"])), _d(x), _d(y), _d(info))
    return x, y, dict(iters=int(info[0]), status=int(info[1]), res_pri=info[2], res_dua=info[3], gap=info[4], polished=int(info[5]))


def polish_generic(qp, x0, actv):
    """Active-set polish of a canonical QP dict from the primal point x0 and the working set actv (int per row): (x, y, info); info["status"] == 1 and
    info["polished"] >= 1 only for a VERIFIED KKT point (oracle/qp.hpp: polish_from)."""
    L = lib()
    n = len(qp["Pd"]); m = len(qp["l"])
    x = np.zeros(n); y = np.zeros(m); info = np.zeros(6)
    Ap = np.ascontiguousarray(qp["Ap"], dtype=np.int32); Ai = np.ascontiguousarray(qp["Ai"], dtype=np.int32); av = np.ascontiguousarray(actv, dtype=np.int32)
    L.po_polish_generic(n, m, _d(_arr(qp["Pd"])), _d(_arr(qp["q"])), Ap.ctypes.data_as(c_ip), Ai.ctypes.data_as(c_ip), _d(_arr(qp["Ax"])),
                        _d(_arr(qp["l"])), _d(_arr(qp["u"])), _d(_arr(x0, n)), av.ctypes.data_as(c_ip), _d(x), _d(y), _d(info))
    return x, y, dict(iters=int(info[0]), status=int(info[1]), res_pri=info[2], res_dua=info[3], gap=info[4], polished=int(info[5]))


def polish_lu(qp, x0, actv, rounds=10):
    """The same active-set polish in scipy (sparse LU of the UNREGULARISED KKT matrix of the working set + iterative refinement): for the long-horizon QPs whose optimum
    carries states of 1e4 -- vertices: as many active rows as variables -- where the regularised LDL' of qp.hpp's polish does not converge.  Accepts, like it, only a KKT
    point of the full QP: every inactive row feasible and every active multiplier non-negative to 1e-9 of the problem's scale, KKT residual of the working set below
    that too.  Returns (x, y, info)."""
    import scipy.sparse as sp, scipy.sparse.linalg as spl
    n = len(qp["Pd"]); m = len(qp["l"])
    A = sp.csc_matrix((qp["Ax"], qp["Ai"], qp["Ap"]), shape=(m, n)).tocsr()
    l, u = np.asarray(qp["l"]), np.asarray(qp["u"])
    eq = l == u; lo = (l > -1e19) & ~eq; up = (u < 1e19) & ~eq; ineq = lo | up
    bnd = np.where(eq | lo, l, u)
    act = (np.asarray(actv) != 0) & ineq | eq
    info = dict(iters=0, status=-2, res_pri=0.0, res_dua=0.0, gap=0.0, polished=0)
    x = np.array(x0, dtype=float); y = np.zeros(m)
    for rnd in range(rounds):
        idx = np.where(act)[0]
        Aa = A[idx]
        K = sp.bmat([[sp.diags(qp["Pd"]), Aa.T], [Aa, None]], format="csc")
        rhs = np.concatenate([-np.asarray(qp["q"]), bnd[idx]])
        try:
            lu = spl.splu((K + sp.diags(np.concatenate([np.full(n, 1e-12), np.full(len(idx), -1e-12)]))).tocsc())
        except RuntimeError:
            return x, y, info
        sol = lu.solve(rhs)
        for _ in range(8):
            d = lu.solve(rhs - K @ sol); sol = sol + d
            if np.max(np.abs(d)) <= 1e-15 * max(1.0, np.max(np.abs(sol))): break
        x = sol[:n]; y = np.zeros(m); y[idx] = sol[n:]
        Ax = A @ x
        scale = 1.0 + max(np.max(np.abs(Ax)), np.max(np.abs(qp["Pd"] * x)), np.max(np.abs(A.T @ y)), np.max(np.abs(qp["q"])))
        res = float(np.max(np.abs(rhs - K @ sol)))
        t = np.where(lo, Ax - l, np.where(up, u - Ax, 0.0)); lam = np.where(lo, -y, np.where(up, y, 0.0))
        drop = ineq & act & (lam < 0.0); add = ineq & ~act & (t < -1e-9 * scale)
        info["res_dua"] = res
        if not np.isfinite(res) or res > 1e-9 * scale: return x, y, info
        if not drop.any() and not add.any():
            info.update(status=1, polished=1 + rnd); return x, y, info
        act = (act & ~drop) | add
    return x, y, info


def active_set(qp, x, y, tol=1e-7):
    """Index list of active inequality rows from an (x, y) pair of the canonical QP: +(i+1) upper-active, -(i+1) lower-active.
    Equality rows (l == u) are excluded.  A row is active when its multiplier exceeds `tol` in magnitude."""
    out = []
    for i in range(len(y)):
        if qp["l"][i] == qp["u"][i]:
            continue
        if y[i] > tol:
            out.append(i + 1)
        elif y[i] < -tol:
            out.append(-(i + 1))
    return out


class OracleDecoupled:
    """Decoupled (lateral) formulation: /root/reference/src/decoupled_lat_long.jl.  Same caveats as Oracle (parity unpinned)."""

    def __init__(self, N_short=10, N_long=20, dt_short=0.01, dt_long=0.2, use_correction_step=True):
        self.L = lib()
        self.L.pd_create.restype = C.c_void_p
        self.L.pd_create.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, C.c_int]
        self.h = C.c_void_p(self.L.pd_create(N_short, N_long, dt_short, dt_long, int(use_correction_step)))
        self.Ns, self.Nl, self.N, self.Nn = N_short, N_long, N_short + N_long, N_short + N_long + 1
        n = C.c_int(); m = C.c_int(); nnz = C.c_int(); sl = C.c_int()
        self.L.pd_qp_dims(self.h, C.byref(n), C.byref(m), C.byref(nnz), C.byref(sl))
        self.n, self.m, self.nnz, self.sd_len = n.value, m.value, nnz.value, sl.value
        cp = np.zeros(11); self.L.pd_get_control_params(self.h, _d(cp))
        self.cp = dict(zip(["V_min", "V_max", "k_V", "k_s", "deltadot_max", "Q_dpsi", "Q_e", "W_beta", "W_r", "R_delta", "R_ddelta"], cp))

    def __del__(self):
        try:
            self.L.pd_destroy(self.h)
        except Exception:
            pass

    def set_trajectory(self, traj12):
        a = _arr(traj12); self.L.pd_set_trajectory(self.h, a.shape[1], _d(a))

    def time_steps(self, t0):
        ts = np.zeros(self.Nn); dt = np.zeros(self.N)
        self.L.pd_time_steps(self.h, C.c_double(t0), _d(ts), _d(dt)); return ts, dt

    def set_time_grid_naive(self, naive):
        self.L.pd_set_time_grid_naive(self.h, int(bool(naive)))

    def simulate_times(self, dt, t_end, steps, t_start=0.0):
        out = np.zeros(steps); self.L.po_simulate_times.restype = C.c_int
        self.L.po_simulate_times(C.c_double(dt), C.c_double(t_end), C.c_double(t_start), int(steps), _d(out))
        return out

    def nodes(self, state6, control3, ts, dt, time_offset=float("nan")):
        qs = np.zeros((self.Nn, 4)); us = np.zeros((self.Nn, 2)); ps = np.zeros((self.Nn, 4))
        self.L.pd_nodes(self.h, _d(_arr(state6, 6)), _d(_arr(control3, 3)), C.c_double(time_offset), _d(_arr(ts, self.Nn)), _d(_arr(dt, self.N)), _d(qs), _d(us), _d(ps))
        return qs, us, ps

    def node_edges(self, state6, control3, ts, dt, time_offset=float("nan")):
        """(edge_L, edge_R) of the tube at every linearization node [Nn, 2]"""
        e = np.zeros((self.Nn, 2))
        self.L.pd_node_edges(self.h, _d(_arr(state6, 6)), _d(_arr(control3, 3)), C.c_double(time_offset), _d(_arr(ts, self.Nn)), _d(_arr(dt, self.N)), _d(e))
        return e

    def update_qp(self, qs, us, ps, dt):
        sd = np.zeros(self.sd_len)
        self.L.pd_update_qp(self.h, _d(_arr(qs, 4 * self.Nn)), _d(_arr(us, 2 * self.Nn)), _d(_arr(ps, 4 * self.Nn)), _d(_arr(dt, self.N)), _d(sd))
        return sd

    def unpack_sd(self, sd):
        N = self.N; o = 0; out = {}
        for name, sz, shp in [("A", 16, (N, 4, 4)), ("B0", 4, (N, 4)), ("Bf", 4, (N, 4)), ("c", 4, (N, 4)), ("H", 8, (N, 4, 2)), ("G", 4, (N, 4)),
                              ("dmin", 1, (N,)), ("dmax", 1, (N,)), ("ddmin", 1, (N,)), ("ddmax", 1, (N,)), ("dt", 1, (N,))]:
            out[name] = sd[o:o + sz * N].reshape(shp); o += sz * N
        out["q_curr"] = sd[o:o + 4]; out["d_curr"] = sd[o + 4]
        return out

    def assemble_qp(self, sd):
        Pd = np.zeros(self.n); q = np.zeros(self.n); Ap = np.zeros(self.n + 1, dtype=np.int32); Ai = np.zeros(self.nnz, dtype=np.int32)
        Ax = np.zeros(self.nnz); l = np.zeros(self.m); u = np.zeros(self.m)
        self.L.pd_assemble_qp(self.h, _d(_arr(sd, self.sd_len)), _d(Pd), _d(q), Ap.ctypes.data_as(c_ip), Ai.ctypes.data_as(c_ip), _d(Ax), _d(l), _d(u))
        return dict(Pd=Pd, q=q, Ap=Ap, Ai=Ai, Ax=Ax, l=l, u=u)

    def solve_exact(self, sd):
        x = np.zeros(self.n); y = np.zeros(self.m); info = np.zeros(6)
        self.L.pd_solve_exact(self.h, _d(_arr(sd, self.sd_len)), _d(x), _d(y), _d(info))
        return x, y, dict(iters=int(info[0]), status=int(info[1]), res_pri=info[2], res_dua=info[3], gap=info[4], polished=int(info[5]))

    def solve_exact_verified(self, sd, qp=None, walls=None, wall_weight=1000.0):
        """Exact optimum as a VERIFIED KKT point of the canonical QP: (x, y, info) with info["polished"] >= 1, or info["status"] != 1 when nothing verifies.
        `qp`: the canonical dict to solve (default assemble_qp(sd); pass the wall-extended dict together with walls = edges[N, 2] of nodes 2..N+1).  First the sparse
        interior point + polish of oracle/qp.hpp; where that does not end in a verified point (long horizons whose optimum diverges), the stage-structured interior
        point of oracle/lat_ipm_numpy.py supplies the candidate and the same canonical polish verifies it."""
        from . import lat_ipm_numpy as lp
        if qp is None:
            x, y, info = self.solve_exact(sd)
            qp = None if (info["status"] == 1 and info["polished"] >= 1) else self.assemble_qp(sd)
        else:
            x, y, info = solve_exact_generic(qp)
            if info["status"] == 1 and info["polished"] >= 1: qp = None
        if qp is None:
            info["method"] = "sparse"; return x, y, info
        D = lp.stage_data(self.unpack_sd(sd), self.cp, walls=walls, wall_weight=wall_weight)
        r = lp.solve(D)
        x0, actv = lp.canonical_candidate(D, r, self.Ns, walls=walls is not None)
        x2, y2, info2 = polish_generic(qp, x0, actv)
        info2["method"] = "stage"; info2["iters"] = r["iters"]
        if info2["status"] == 1: return x2, y2, info2
        x2, y2, info2 = polish_lu(qp, x0, actv)
        info2["method"] = "stage+lu"; info2["iters"] = r["iters"]
        if info2["status"] == 1: return x2, y2, info2
        info["method"] = "none"; info["status"] = info["status"] if info["status"] != 1 else -2
        return x, y, info

    def step_batch(self, states6, controls3, t0, time_offsets=None, nthreads=1):
        """Whole lateral step per instance with the OSQP port (cold): (u [B,3], iters, status, wall seconds)."""
        B = len(t0)
        u = np.zeros((B, 3)); it = np.zeros(B, dtype=np.int32); st = np.zeros(B, dtype=np.int32)
        to = _arr(time_offsets, B) if time_offsets is not None else None
        self.L.pd_step_batch.restype = C.c_double
        secs = self.L.pd_step_batch(self.h, B, _d(_arr(states6, 6 * B)), _d(_arr(controls3, 3 * B)), _d(_arr(t0, B)), _d(to) if to is not None else None, nthreads,
                                    _d(u), it.ctypes.data_as(c_ip), st.ctypes.data_as(c_ip))
        return u, it, st, secs

    def split_x(self, x):
        Nn, N = self.Nn, self.N
        return dict(q=x[:4 * Nn].reshape(Nn, 4), delta=x[4 * Nn:5 * Nn], sigma=x[5 * Nn:5 * Nn + 2 * N].reshape(N, 2), ddelta=x[5 * Nn + 2 * N:])

    def lateral_dynamics(self, q4, u2, p4):
        o = np.zeros(4); self.L.pd_lateral_dynamics(self.h, _d(_arr(q4, 4)), _d(_arr(u2, 2)), _d(_arr(p4, 4)), _d(o)); return o

    def linearize_interval(self, q4, w0, wf, dt, ramp):
        A = np.zeros((4, 4)); B0 = np.zeros(4); Bf = np.zeros(4); c = np.zeros(4)
        self.L.pd_linearize_interval(self.h, _d(_arr(q4, 4)), _d(_arr(w0, 6)), _d(_arr(wf, 6)), C.c_double(dt), int(ramp), _d(A), _d(B0), _d(Bf), _d(c))
        return A, B0, Bf, c

    def next_control(self, delta, Fx_seed):
        o = np.zeros(3); self.L.pd_next_control(self.h, C.c_double(delta), C.c_double(Fx_seed), _d(o)); return o


# ---- glue between the lateral oracle and the product's embedded QP layout (pg_get_qp of a PG_DECOUPLED handle); used by tests/ and by bench.py's accuracy block ----

def embed_sd(orc, sd, ux_dummy=8.0):
    """Oracle stage data of the lateral QP -> the embedded coupled layout pg_get_qp returns for PG_DECOUPLED handles."""
    S = orc.unpack_sd(sd); N = orc.N
    A = np.zeros((N, 6, 6)); A[:, 0, 0] = 1; A[:, 1, 1] = 1; A[:, 2:, 2:] = S["A"]
    B0 = np.zeros((N, 6, 2)); B0[:, 2:, 0] = S["B0"]; Bf = np.zeros((N, 6, 2)); Bf[:, 2:, 0] = S["Bf"]
    c = np.zeros((N, 6)); c[:, 2:] = S["c"]
    return np.concatenate([A.ravel(), B0.ravel(), Bf.ravel(), c.ravel(), S["H"].ravel(), S["G"].ravel(), S["dmin"], S["dmax"], np.ones(N), S["ddmin"], S["ddmax"],
                           S["dt"], [0.0, ux_dummy], S["q_curr"], [S["d_curr"], 0.0], [0.0, 0.0], [1.0]])


def unembed_qp(orc, row):
    """Inverse of embed_sd: one row of pg_get_qp of a PG_DECOUPLED handle -> the oracle's lateral stage data."""
    N = orc.N; o = 0
    def take(n, shape):
        nonlocal o
        v = row[o:o + n].reshape(shape); o += n
        return v
    A = take(36 * N, (N, 6, 6)); B0 = take(12 * N, (N, 6, 2)); Bf = take(12 * N, (N, 6, 2)); c = take(6 * N, (N, 6))
    H = take(8 * N, (N, 4, 2)); G = take(4 * N, (N, 4)); dmin = take(N, (N,)); dmax = take(N, (N,)); take(N, (N,)); ddmin = take(N, (N,)); ddmax = take(N, (N,)); dt = take(N, (N,))
    qc = take(6, (6,)); uc = take(2, (2,))
    return np.concatenate([A[:, 2:, 2:].ravel(), B0[:, 2:, 0].ravel(), Bf[:, 2:, 0].ravel(), c[:, 2:].ravel(), H.ravel(), G.ravel(), dmin, dmax, ddmin, ddmax, dt, qc[2:], [uc[0]]])


def extend_with_walls(orc, qpc, edges, dt, Ww):
    """Canonical lateral QP + the build-defined wall rows (columns n..n+N-1 = sw_k; rows m+3k: e - sw <= edge_L, m+3k+1: e + sw >= edge_R, m+3k+2: sw >= 0)."""
    n, m, Nh = orc.n, orc.m, orc.N
    A = sp.csc_matrix((qpc["Ax"], qpc["Ai"], qpc["Ap"]), shape=(m, n))
    k = np.arange(Nh); col = 4 * (k + 1) + 3
    rows = np.concatenate([3 * k, 3 * k, 3 * k + 1, 3 * k + 1, 3 * k + 2]); cols = np.concatenate([col, n + k, col, n + k, n + k])
    vals = np.concatenate([np.ones(Nh), -np.ones(Nh), np.ones(Nh), np.ones(Nh), np.ones(Nh)])
    W = sp.csc_matrix((vals, (rows, cols)), shape=(3 * Nh, n + Nh))
    lw = np.full(3 * Nh, -1e20); uw = np.full(3 * Nh, 1e20)
    uw[3 * k] = edges[:, 0]; lw[3 * k + 1] = edges[:, 1]; lw[3 * k + 2] = 0.0
    Aw = sp.vstack([sp.hstack([A, sp.csc_matrix((m, Nh))]), W]).tocsc(); Aw.sort_indices()
    return dict(Pd=np.concatenate([qpc["Pd"], np.zeros(Nh)]), q=np.concatenate([qpc["q"], Ww * dt]), Ap=Aw.indptr, Ai=Aw.indices, Ax=Aw.data,
                l=np.concatenate([qpc["l"], lw]), u=np.concatenate([qpc["u"], uw])), Aw
