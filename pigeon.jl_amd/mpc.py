"""Host-side mirror of the reference's MPC call convention for a BATCH of independent instances.

Reference seam (relative to /root/reference/src): the five generic functions of model_predictive_control.jl:70-78
applied to a mutable TrajectoryTrackingMPC (:32-68), called in this order by ros_integration.jl:96-99,124 and by
simulate (:80-100).  Julia's `f!(mpc)` spelling becomes `f_(mpc)` here; argument meaning and order are unchanged.
Everything numerical happens in libpigeon_hip.so (HIP, gfx950); this module only marshals arrays.
"""
import ctypes as C
import math

import numpy as np

from . import _lib
from .trajectories import TrajectoryTube
from .vehicles import X1, CoupledControlParams, DecoupledControlParams

c_dp = C.POINTER(C.c_double)

SOLVED, MAX_ITER, NUMERICAL, INFEASIBLE_X0, SOLVED_UNVERIFIED = 1, 2, 3, 4, 5


def is_solved(status):
    """True where the solver returned an answer that met its tolerances: PG_SOLVED (with the polish on: a VERIFIED KKT point) or PG_SOLVED_UNVERIFIED (the
    interior-point iterate no active-set round could verify; include/pigeon_mpc.h)."""
    status = np.asarray(status)
    return (status == SOLVED) | (status == SOLVED_UNVERIFIED)


def _p(a, ctype=c_dp):
    return None if a is None else a.ctypes.data_as(ctype)


def _f64(x, shape=None):
    a = np.ascontiguousarray(x, dtype=np.float64)
    if shape is not None:
        a = a.reshape(shape)
    return a


class BatchedTrajectoryTrackingMPC:
    """B copies of CoupledTrajectoryTrackingMPC(vehicle, trajectory; ...) (coupled_lat_long.jl:42-60) on one MI355X."""

    def __init__(self, trajectory, batch_capacity, vehicle=None, control_params=None, N_short=10, N_long=20, dt_short=0.01, dt_long=0.2,
                 use_correction_step=True, rk4_substeps=10, device=0, ipm_max_iter=40, ipm_tol=None, ipm_mu0=100.0, hji_eps=0.05, formulation="coupled",
                 precision="f64", walls=False, wall_weight=1000.0, polish=None, polish_rho=None, polish_tol=None, polish_ipm_tol=None, warm_polish=None, cold_guess=None,
                 options=None, phase_timing=True, allow_f32_long_lateral=False):
        """options: {name: value} of build-defined options applied right after pg_create (pg_set_option, include/pigeon_mpc.h); precision "f64-diag" loads the
        diagnostic build of the fp64 library (tests / tools only).  phase_timing: this mirror is the test / bench harness and switches the library's per-phase HIP events ON
        by default (phase_ms(); the library's own default is off -- 2-4 % of a step: bench.py times its loops with phase_timing=False)."""
        self.precision = precision
        self.real = np.float32 if precision == "f32" else np.float64      # element type of DEVICE arrays handed to the *_dev entry points
        self.lib = _lib.load_library(precision)
        cfg = _lib.pg_config()
        assert formulation in ("coupled", "decoupled")
        self.formulation = formulation
        (self.lib.pg_default_config if formulation == "coupled" else self.lib.pg_default_config_decoupled)(C.byref(cfg))
        # solver tolerances default to the library's own (they depend on its arithmetic type: pg_default_config*)
        ipm_tol = cfg.ipm_tol if ipm_tol is None else ipm_tol
        if polish is not None:                    # None: the library's default (on for both formulations)
            cfg.polish = int(bool(polish))
        if polish_rho is not None:
            cfg.polish_rho = float(polish_rho)
        if polish_tol is not None:
            cfg.polish_tol = float(polish_tol)
        if polish_ipm_tol is not None:
            cfg.polish_ipm_tol = float(polish_ipm_tol)
        if warm_polish is not None:
            cfg.warm_polish = int(bool(warm_polish))
        if cold_guess is not None:
            cfg.cold_guess = int(cold_guess)
        self.vehicle = X1() if vehicle is None else dict(vehicle)
        default_cp = CoupledControlParams() if formulation == "coupled" else DecoupledControlParams()
        self.control_params = default_cp if control_params is None else dict(control_params)
        for name, _ in _lib.pg_vehicle._fields_:
            setattr(cfg.vehicle, name, float(self.vehicle[name]))
        for name, _ in _lib.pg_control_params._fields_:
            if name == "_pad" or name not in self.control_params:      # the lateral formulation has no Q_ds / R_Fx / R_dFx / W_HJI / N_HJI
                continue
            setattr(cfg.control, name, int(self.control_params[name]) if name == "N_HJI" else float(self.control_params[name]))
        cfg.N_short, cfg.N_long, cfg.dt_short, cfg.dt_long = N_short, N_long, dt_short, dt_long
        cfg.use_correction_step, cfg.rk4_substeps, cfg.batch_capacity, cfg.device = int(use_correction_step), rk4_substeps, batch_capacity, device
        cfg.ipm_max_iter, cfg.ipm_tol, cfg.ipm_mu0, cfg.hji_eps = ipm_max_iter, ipm_tol, ipm_mu0, hji_eps
        cfg.walls = int(bool(walls))          # build-defined extension: soft rows edge_R - sw <= e <= edge_L + sw (decoupled formulation)
        cfg.wall_weight = float(wall_weight)
        cfg.allow_f32_long_lateral = int(bool(allow_f32_long_lateral))      # (the fp32 library refuses the decoupled formulation beyond 32 intervals without it: pigeon_mpc.h)
        self.wall_weight = float(wall_weight)
        self.walls = bool(walls)
        self.cfg = cfg
        self.h = C.c_void_p()
        rc = self.lib.pg_create(C.byref(cfg), C.byref(self.h))
        if rc != 0:
            raise _lib.PigeonError(f"pg_create failed with status {rc}: {self.lib.pg_last_error(None).decode()}")
        self.N_short, self.N_long = N_short, N_long
        self.N = N_short + N_long
        self.NN = self.N + 1
        self.capacity = batch_capacity
        self.B = 0
        un = np.zeros(2)
        self._chk(self.lib.pg_get_u_normalization(self.h, _p(un)), "pg_get_u_normalization")
        self.u_normalization = un
        self.qp_len = self.lib.pg_qp_len(self.h)
        self.trajectory = None
        try:
            self.set_option("phase_timing", 1 if phase_timing else 0)
        except _lib.PigeonError:          # (an older build selected through PIGEON_HIP_LIB for an A/B run: its events are always on)
            pass
        for name, value in (options or {}).items():
            self.set_option(name, value)
        if trajectory is not None:
            self.set_trajectory(trajectory)

    def _chk(self, rc, what):
        _lib.check(self.lib, self.h, rc, what)

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.lib.pg_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- mpc.trajectory = ... (ros_integration.jl:53) ----
    def set_trajectory(self, traj):
        """One TrajectoryTube for the whole batch, or a list of tubes (a library; select per instance with set_trajectory_index)."""
        if isinstance(traj, (list, tuple)):
            return self.set_trajectories(traj)
        self.trajectory = traj
        self.trajectories = [traj]
        cols = [np.ascontiguousarray(traj.data[i]) for i in range(12)]
        self._chk(self.lib.pg_set_trajectory(self.h, len(traj), *[_p(c) for c in cols]), "pg_set_trajectory")

    # ---- one controller per (x0, reference trajectory) pair: the batch carries a library of tubes and a per-instance selection ----
    def set_trajectories(self, trajs, index=None):
        trajs = list(trajs)
        Lmax = max(len(t) for t in trajs)
        L = np.array([len(t) for t in trajs], dtype=np.int32)
        pack = np.zeros((len(trajs), 10, Lmax))
        rows = [0, 1, 2, 3, 4, 5, 6, 7, 10, 11]                     # t, s, V, A, E, N, psi, kappa, edge_L, edge_R of the 12-channel tube
        for k, t in enumerate(trajs):
            pack[k, :, :len(t)] = t.data[rows]
            pack[k, :, len(t):] = t.data[rows][:, -1:]              # padding is never read (L[k] bounds every search)
        self._chk(self.lib.pg_set_trajectories(self.h, len(trajs), Lmax, _p(L, C.POINTER(C.c_int32)), _p(pack)), "pg_set_trajectories")
        self.trajectories = trajs
        self.trajectory = trajs[0]
        if index is not None:
            self.set_trajectory_index(index)

    def set_trajectory_index(self, index):
        index = np.ascontiguousarray(index, dtype=np.int32)
        self._chk(self.lib.pg_set_trajectory_index(self.h, len(index), _p(index, C.POINTER(C.c_int32))), "pg_set_trajectory_index")
        self.trajectory_index = index

    # ---- mpc.HJI_cache = HJICache(...) (Pigeon.jl:40) ----
    def set_hji_cache(self, grid_knots, V_raw, gradV_raw):
        dims = np.array([len(k) for k in grid_knots], dtype=np.int32)
        kc = np.ascontiguousarray(np.concatenate([np.asarray(k, dtype=np.float32) for k in grid_knots]))
        V = np.ascontiguousarray(V_raw, dtype=np.float32).reshape(-1)
        g = np.ascontiguousarray(gradV_raw, dtype=np.float32).reshape(-1)
        assert V.size == int(np.prod(dims.astype(np.int64))) and g.size == 7 * V.size
        self._chk(self.lib.pg_set_hji_grid(self.h, _p(dims, C.POINTER(C.c_int32)), _p(kc, C.POINTER(C.c_float)), _p(V, C.POINTER(C.c_float)),
                                           _p(g, C.POINTER(C.c_float))), "pg_set_hji_grid")

    def clear_hji_cache(self):
        self._chk(self.lib.pg_clear_hji_grid(self.h), "pg_clear_hji_grid")

    # ---- mpc.solved = false (ros_integration.jl:34,41,147) ----
    def reset(self, mask=None):
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        self._chk(self.lib.pg_reset(self.h, _p(m, C.POINTER(C.c_uint8))), "pg_reset")

    # ---- mpc.current_state / current_control / other_car_state / time_offset + the time argument of compute_time_steps! ----
    def set_inputs(self, current_state, current_control, t, other_car_state=None, time_offset=None):
        s = _f64(current_state).reshape(-1, 6)
        B = s.shape[0]
        c = _f64(current_control, (B, 3)); t0 = _f64(t, (B,))
        o = None if other_car_state is None else _f64(other_car_state, (B, 4))
        to = None if time_offset is None else _f64(time_offset, (B,))
        self._chk(self.lib.pg_set_inputs(self.h, B, _p(s), _p(c), _p(t0), _p(o), _p(to)), "pg_set_inputs")
        self.B = B

    def set_inputs_dev(self, B, state_ptr, control_ptr, t0_ptr, other_ptr=None, toff_ptr=None):
        """Inputs already resident in HBM (raw device addresses, e.g. torch.Tensor.data_ptr())."""
        vp = lambda a: C.c_void_p(a) if a else None
        self._chk(self.lib.pg_set_inputs_dev(self.h, B, vp(state_ptr), vp(control_ptr), vp(t0_ptr), vp(other_ptr), vp(toff_ptr)), "pg_set_inputs_dev")
        self.B = B

    # ---- the five reference calls ----
    def compute_time_steps_(self, t0=None):
        """compute_time_steps!(mpc, t0): model_predictive_control.jl:70.  t0 (array of B) may also come from set_inputs."""
        if t0 is not None:
            raise ValueError("pass t through set_inputs(state, control, t, ...): the batch keeps its inputs in device memory")
        self._chk(self.lib.pg_compute_time_steps(self.h), "pg_compute_time_steps")

    def compute_linearization_nodes_(self):
        self._chk(self.lib.pg_compute_linearization_nodes(self.h), "pg_compute_linearization_nodes")

    def update_QP_(self):
        self._chk(self.lib.pg_update_qp(self.h), "pg_update_qp")

    def solve_(self):
        self._chk(self.lib.pg_solve(self.h), "pg_solve")

    def get_next_control(self):
        """BicycleControl (delta, Fxf, Fxr) per instance: coupled_lat_long.jl:370-374."""
        u = np.zeros((self.B, 3))
        self._chk(self.lib.pg_get_next_control(self.h, _p(u)), "pg_get_next_control")
        return u

    def get_next_control_hji(self, use_HJI_policy=True):
        """ros_integration.jl:114-124: (u [B,3], source [B] (0 MPC / 1 HJI policy / 2 unsafe but policy off), optimal_control (delta, Fx) [B,2])"""
        u = np.zeros((self.B, 3)); src = np.zeros(self.B, dtype=np.int32); u2 = np.zeros((self.B, 2))
        self._chk(self.lib.pg_get_next_control_hji(self.h, int(bool(use_HJI_policy)), _p(u), _p(src, C.POINTER(C.c_int32)), _p(u2)), "pg_get_next_control_hji")
        return u, src, u2

    def step_(self, current_state, current_control, t, other_car_state=None, time_offset=None):
        """The whole callback body of ros_integration.jl:94-99,124 for every instance; returns (u, status, iters)."""
        s = _f64(current_state).reshape(-1, 6); B = s.shape[0]
        c = _f64(current_control, (B, 3)); t0 = _f64(t, (B,))
        o = None if other_car_state is None else _f64(other_car_state, (B, 4))
        to = None if time_offset is None else _f64(time_offset, (B,))
        u = np.zeros((B, 3)); st = np.zeros(B, dtype=np.int32); it = np.zeros(B, dtype=np.int32)
        self._chk(self.lib.pg_step(self.h, B, _p(s), _p(c), _p(t0), _p(o), _p(to), _p(u), _p(st, C.POINTER(C.c_int32)), _p(it, C.POINTER(C.c_int32))), "pg_step")
        self.B = B
        return u, st, it

    MPC_ = step_      # the convenience names BASELINE.json's north star uses

    def step_dev(self, u_out_ptr=None):
        self._chk(self.lib.pg_step_dev(self.h, C.c_void_p(u_out_ptr) if u_out_ptr else None), "pg_step_dev")

    def simulate_(self, steps, dt=0.01, record=False):
        """simulate (model_predictive_control.jl:80-100) on the device from the inputs last installed; returns (state, control, t) after `steps`
        steps and, with record=True, the histories qs [steps][B][6], us [steps][B][3] (the values pushed at :88-89)."""
        import ctypes as C_
        qh = uh = None; dq = du = None
        if record:
            import torch
            tdt = torch.float32 if self.precision == "f32" else torch.float64         # device records have the library's own element type
            dq = torch.empty(steps, self.B, 6, dtype=tdt, device=f"cuda:{self.cfg.device}"); du = torch.empty(steps, self.B, 3, dtype=tdt, device=f"cuda:{self.cfg.device}")
        self._chk(self.lib.pg_simulate_dev(self.h, steps, C_.c_double(dt), C_.c_void_p(dq.data_ptr()) if record else None, C_.c_void_p(du.data_ptr()) if record else None), "pg_simulate_dev")
        s = np.zeros((self.B, 6)); c = np.zeros((self.B, 3)); t = np.zeros(self.B)
        self._chk(self.lib.pg_get_state(self.h, _p(s), _p(c), _p(t)), "pg_get_state")
        if record:
            qh = dq.cpu().numpy().astype(np.float64); uh = du.cpu().numpy().astype(np.float64)
        return s, c, t, qh, uh

    def simulate_clock(self, steps, t_start, dt=0.01):
        """The times the rollout's loop variable takes, per instance: (t_start .+ (0:dt:trajectory.t[end]))[1:steps] as Julia's range arithmetic gives them
        (model_predictive_control.jl:87; pg_simulate_clock) -- [steps][B]."""
        import ctypes as C_
        ts = _f64(t_start).reshape(-1).copy(); out = np.zeros((steps, len(ts)))
        self._chk(self.lib.pg_simulate_clock(self.h, C_.c_double(dt), int(steps), len(ts), _p(ts), _p(out)), "pg_simulate_clock")
        return out

    def synchronize(self):
        self._chk(self.lib.pg_synchronize(self.h), "pg_synchronize")

    def set_stream(self, hip_stream):
        self._chk(self.lib.pg_set_stream(self.h, C.c_void_p(hip_stream)), "pg_set_stream")

    def set_fusion(self, mode):
        """Fused step (update_QP! inside the solve kernel, include/pigeon_mpc.h pg_set_fusion): False / 0 never (the default), True / 1 always, 2 for all-warm batches."""
        self._chk(self.lib.pg_set_fusion(self.h, int(mode)), "pg_set_fusion")

    def set_pipeline(self, mode):
        """Pipelined nodes + update_QP launch for large batches with cold instances (include/pigeon_mpc.h pg_set_pipeline): True / 1 where it applies (default), False / 0 never."""
        self._chk(self.lib.pg_set_pipeline(self.h, int(mode)), "pg_set_pipeline")

    def set_option(self, name, value):
        """Build-defined option of this handle by name (include/pigeon_mpc.h pg_set_option: solver rules, launch shape, lateral-solver tuning)."""
        self._chk(self.lib.pg_set_option(self.h, name.encode(), C.c_double(float(value))), f"pg_set_option({name})")

    def get_option(self, name):
        """Current value of an option, or a read-only launch statistic ("stat_pipelined_launches", ...)."""
        v = C.c_double(0.0)
        self._chk(self.lib.pg_get_option(self.h, name.encode(), C.byref(v)), f"pg_get_option({name})")
        return v.value

    def pipeline_fallbacks(self):
        """Cumulative number of waiting wavefronts of the pipelined nodes + update_QP launch that gave up (the step was then redone launch per phase)."""
        n = C.c_int64(0)
        self._chk(self.lib.pg_get_pipeline_fallbacks(self.h, C.byref(n)), "pg_get_pipeline_fallbacks")
        return int(n.value)

    def phase_ms(self):
        out = (C.c_float * 3)()
        self._chk(self.lib.pg_get_phase_ms(self.h, out), "pg_get_phase_ms")
        return [out[i] for i in range(3)]

    # ---- read-backs ----
    def time_steps(self):
        ts = np.zeros((self.B, self.NN)); dt = np.zeros((self.B, self.N)); pts = np.zeros((self.B, self.NN))
        self._chk(self.lib.pg_get_time_steps(self.h, _p(ts), _p(dt), _p(pts)), "pg_get_time_steps")
        return ts, dt, pts

    def nodes(self):
        qs = np.zeros((self.B, self.NN, 6)); us = np.zeros((self.B, self.NN, 2)); ps = np.zeros((self.B, self.NN, 4))
        self._chk(self.lib.pg_get_nodes(self.h, _p(qs), _p(us), _p(ps)), "pg_get_nodes")
        return qs, us, ps

    def path_coordinates(self):
        sep = np.zeros((self.B, 3))
        self._chk(self.lib.pg_get_path_coordinates(self.h, _p(sep)), "pg_get_path_coordinates")
        return sep

    def qp_data(self, b0=0, n=None):
        n = self.B - b0 if n is None else n
        out = np.zeros((n, self.qp_len))
        self._chk(self.lib.pg_get_qp(self.h, b0, n, _p(out)), "pg_get_qp")
        return out

    def set_qp_data(self, qp, b0=0):
        """Install QP data (layout of qp_data) for instances [b0, b0 + len(qp)): solve_() then solves exactly these problems (include/pigeon_mpc.h pg_set_qp)."""
        q = _f64(qp).reshape(-1, self.qp_len)
        self._chk(self.lib.pg_set_qp(self.h, b0, q.shape[0], _p(q)), "pg_set_qp")

    def solution(self):
        x = np.zeros((self.B, self.NN, 8)); sg = np.zeros((self.B, self.N, 3))
        self._chk(self.lib.pg_get_solution(self.h, _p(x), _p(sg)), "pg_get_solution")
        return x, sg

    def solve_info(self):
        st = np.zeros(self.B, dtype=np.int32); it = np.zeros(self.B, dtype=np.int32); act = np.zeros((self.B, self.N), dtype=np.uint16); mu = np.zeros(self.B)
        self._chk(self.lib.pg_get_solve_info(self.h, _p(st, C.POINTER(C.c_int32)), _p(it, C.POINTER(C.c_int32)), _p(act, C.POINTER(C.c_uint16)), _p(mu)),
                  "pg_get_solve_info")
        return st, it, act, mu

    def polish_info(self):
        """Per instance: 0 polish not run, k >= 1 verified in round k, -1 not verified (interior-point iterate kept)."""
        p = np.zeros(self.B, dtype=np.int32)
        self._chk(self.lib.pg_get_polish_info(self.h, _p(p, C.POINTER(C.c_int32))), "pg_get_polish_info")
        return p

    def multipliers(self):
        """Multipliers of the inequality rows of the last solve [B, N, 16], indexed like the bits of the active masks (include/pigeon_mpc.h pg_get_multipliers)."""
        lam = np.zeros((self.B, self.N, 16))
        self._chk(self.lib.pg_get_multipliers(self.h, _p(lam)), "pg_get_multipliers")
        return lam

    def hji_constraint(self):
        M = np.zeros((self.B, 2)); b = np.zeros(self.B); V = np.zeros(self.B)
        self._chk(self.lib.pg_get_hji_constraint(self.h, _p(M), _p(b), _p(V)), "pg_get_hji_constraint")
        return M, b, V

    def wall_edges(self):
        """(edge_L, edge_R) at nodes 2..N+1 [B, N, 2] (wall extension)"""
        e = np.zeros((self.B, self.N, 2))
        self._chk(self.lib.pg_get_walls(self.h, _p(e)), "pg_get_walls")
        return e

    def hji_lookup(self, x7):
        """cache[x] for a batch of HJIRelativeState rows: HJI_computation.jl:66-72."""
        x = _f64(x7).reshape(-1, 7); B = x.shape[0]
        V = np.zeros(B); g = np.zeros((B, 7))
        self._chk(self.lib.pg_hji_lookup(self.h, B, _p(x), _p(V), _p(g)), "pg_hji_lookup")
        return V, g

    def hji_value_slice(self, q7, colors=False, contour=True):
        """rviz.jl:23-40,60-69 for a batch of relative states: V [B, n1, n2] at every knot pair of grid dimensions 1, 2 (the other five components from q7),
        optionally the marker colours [B, n1, n2, 3] and the zero-level crossings on the grid edges (cross_x [B, n1-1, n2], cross_y [B, n1, n2-1]; NaN = none)."""
        q = _f64(q7).reshape(-1, 7); B = q.shape[0]
        dims = np.zeros(7, dtype=np.int32)
        self._chk(self.lib.pg_hji_grid_dims(self.h, _p(dims, C.POINTER(C.c_int32))), "pg_hji_grid_dims")
        n1, n2 = int(dims[0]), int(dims[1])
        V = np.zeros((B, n1, n2)); rgb = np.zeros((B, n1, n2, 3)) if colors else None
        cx = np.zeros((B, n1 - 1, n2)) if contour else None; cy = np.zeros((B, n1, n2 - 1)) if contour else None
        self._chk(self.lib.pg_hji_slice(self.h, B, _p(q), _p(V), _p(rgb), _p(cx), _p(cy)), "pg_hji_slice")
        return V, rgb, cx, cy

    # ---- canonical active-set indices (row numbering of the reference's QP) ----
    def canonical_active_set(self, b, act_masks, qp_row, lam=None, tol=1e-6):
        """Signed 1-based row indices of the reference QP that are active for instance b: +i upper bound, -i lower bound.
        act_masks: [N] uint16 from solve_info(); qp_row: this instance's pg_get_qp block (for the fixed first node).
        lam: this instance's multipliers [N, 16] (multipliers()[b]): with them the CANONICAL rule applies -- a row is active when its bit is set AND its multiplier
        exceeds tol, the rule oracle.active_set applies to the oracle's multipliers (rows held with a zero multiplier are degenerate: either side is a KKT point)."""
        N, Ns = self.N, self.N_short
        cp = self.control_params
        r_C1 = 0; r_C2 = r_C1 + 2 * N; r_C3 = r_C2 + Ns; r_C4 = r_C3 + N; r_C5 = r_C4 + N; r_C6 = r_C5 + N + 1; r_C7 = r_C6 + N + 1
        r_C8 = r_C7 + N + 1; r_C9 = r_C8 + 6; r_C10 = r_C9 + 2; r_C11 = r_C10 + 6 * Ns; r_C12 = r_C11 + Ns; r_C13 = r_C12 + 6 * self.N_long
        out = []
        nh = min(int(cp["N_HJI"]), Ns)
        # node 1 (fixed): sigma_HJI_1 = max(0, -(M u_1 + b)); its two rows are decided in closed form
        M = qp_row[-3:-1]; bh = qp_row[-1]; u1 = qp_row[-5:-3]
        if nh >= 1 and cp["W_HJI"] > 0:
            if M @ u1 + bh < 0:
                out.append(-(r_C11 + 0 + 1))
            else:
                out.append(-(r_C2 + 0 + 1))
        for k in range(N):
            m = int(act_masks[k]); node = k + 1
            bit = lambda j: ((m >> j) & 1) and (lam is None or lam[k][j] > tol)
            if bit(0): out.append(-(r_C5 + node + 1))
            if bit(1): out.append(+(r_C6 + node + 1))
            if bit(2): out.append(-(r_C7 + node + 1))
            base = r_C13 + 9 * k
            if bit(3): out.append(+(base + 0 + 1))
            if bit(4): out.append(-(base + 1 + 1))
            if bit(5): out.append(+(base + 2 + 1))
            for i in range(4):
                if bit(6 + i): out.append(+(base + 3 + i + 1))
            if bit(10): out.append(-(r_C1 + 2 * k + 0 + 1))
            if bit(11): out.append(-(r_C1 + 2 * k + 1 + 1))
            if bit(12): out.append(+(base + 7 + 1))
            if bit(13): out.append(-(base + 8 + 1))
            if node < nh:
                if bit(14): out.append(-(r_C11 + node + 1))
                if bit(15): out.append(-(r_C2 + node + 1))
        return sorted(out, key=abs)


def CoupledTrajectoryTrackingMPC(vehicle, trajectory, batch_capacity=1, **kw):
    """Name of the reference constructor (coupled_lat_long.jl:42); returns the batched type."""
    return BatchedTrajectoryTrackingMPC(trajectory, batch_capacity, vehicle=vehicle, **kw)


def DecoupledTrajectoryTrackingMPC(vehicle, trajectory, batch_capacity=1, **kw):
    """Name of the reference constructor (decoupled_lat_long.jl:32); returns the batched type."""
    return BatchedTrajectoryTrackingMPC(trajectory, batch_capacity, vehicle=vehicle, formulation="decoupled", **kw)


def decoupled_canonical_active_set(N, N_short, act_masks, walls=False, lam=None, tol=1e-6):
    """Signed 1-based active rows of the reference's LATERAL QP (decoupled_lat_long.jl:166-211 row order) from the per-stage masks (lam [N, 16]: the canonical rule of
    BatchedTrajectoryTrackingMPC.canonical_active_set -- bit set and multiplier > tol).
    With the wall extension the 3N wall rows are numbered after the reference's rows: (e - sw <= edge_L, e + sw >= edge_R, sw >= 0) per node 2..N+1."""
    Ns, Nl = N_short, N - N_short
    r_1 = 0; r_2 = 2 * N; r_3 = r_2 + N; r_4 = r_3 + 4; r_5 = r_4 + 1; r_6 = r_5 + 4 * Ns; r_7 = r_6 + 4 * Nl
    out = []
    for k in range(N):
        m = int(act_masks[k]); base = r_7 + 8 * k
        bit = lambda j: ((m >> j) & 1) and (lam is None or lam[k][j] > tol)
        if bit(3): out.append(+(base + 0 + 1))
        if bit(4): out.append(-(base + 1 + 1))
        for i in range(4):
            if bit(6 + i): out.append(+(base + 2 + i + 1))
        if bit(10): out.append(-(r_1 + 2 * k + 0 + 1))
        if bit(11): out.append(-(r_1 + 2 * k + 1 + 1))
        if bit(12): out.append(+(base + 6 + 1))
        if bit(13): out.append(-(base + 7 + 1))
        if walls:
            r_8 = r_7 + 8 * N
            if bit(0): out.append(+(r_8 + 3 * k + 0 + 1))
            if bit(1): out.append(-(r_8 + 3 * k + 1 + 1))
            if bit(2): out.append(-(r_8 + 3 * k + 2 + 1))
    return sorted(out, key=abs)


def simulate(mpc: BatchedTrajectoryTrackingMPC, plant_step, q0, u0, steps, dt=0.01, t_start=None, time_offset=None):
    """Closed-loop harness with the semantics of simulate (model_predictive_control.jl:80-100): the state is advanced with the
    OLD control, then the control is replaced (one-step actuation delay).  `plant_step(q[B,6], u[B,3], dt) -> q` is supplied by
    the caller (the nonlinear plant is not part of the GPU hot path)."""
    q = _f64(q0).reshape(-1, 6).copy(); u = _f64(u0, q.shape[:1] + (3,)).copy()
    t = np.zeros(q.shape[0]) if t_start is None else _f64(t_start, (q.shape[0],)).copy()
    hist = []
    clock = mpc.simulate_clock(steps, t, dt)          # `for t in 0:dt:trajectory.t[end]` (:87) is a Julia range, not an accumulation
    for k in range(steps):
        hist.append((q.copy(), u.copy()))
        mpc.set_inputs(q, u, clock[k], time_offset=time_offset)
        mpc.compute_time_steps_(); mpc.compute_linearization_nodes_(); mpc.update_QP_(); mpc.solve_()
        q = plant_step(q, u, dt)
        u = mpc.get_next_control()
    return hist
