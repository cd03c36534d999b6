/*
 * pigeon_mpc.h — C ABI of the MI355X-native batched MPC hot path (libpigeon_hip.so).
 *
 * The reference (StanfordASL/Pigeon.jl) has NO FFI for this path; its seam is a Julia call convention: five generic
 * functions applied in fixed order to a mutable TrajectoryTrackingMPC
 *   compute_time_steps!(mpc, t)          src/model_predictive_control.jl:70   (-> :17-30)
 *   compute_linearization_nodes!(mpc)    src/model_predictive_control.jl:72   (-> src/coupled_lat_long.jl:62-142)
 *   update_QP!(mpc)                      src/model_predictive_control.jl:74   (-> src/coupled_lat_long.jl:315-368)
 *   solve!(mpc)                          src/model_predictive_control.jl:76   (-> Parametron -> OSQP, third-party)
 *   get_next_control(mpc)                src/model_predictive_control.jl:78   (-> src/coupled_lat_long.jl:370-374)
 * called from src/ros_integration.jl:96-99,124 and src/model_predictive_control.jl:90-95.  Each entry point below names
 * the reference interface it replaces.  A batch of B independent MPC instances shares one handle (one vehicle, one
 * set of control parameters, one reference trajectory, one HJI grid); per-instance persistent state (solved flag,
 * previous time grid, previous primal solution) lives in device memory inside the handle.
 *
 * Conventions: every function returns 0 on success or a negative pg_status; no exceptions cross the ABI; all
 * arrays are instance-major ("array of structs": state[b*6 + k]) in double precision; pointers are HOST pointers
 * unless the parameter name ends in _dev.  One handle per host thread and device.
 *
 * Two builds export this same ABI: libpigeon_hip.so computes the path in fp64 (the reference's type), libpigeon_hip_f32.so in fp32
 * (BASELINE configs 3/4).  Host arrays are double in both; DEVICE arrays of path data (pg_real_dev) have the library's own element
 * type, which pg_precision_bits() reports (64 or 32).  Absolute times (t0, time_offset, the time grid) are double in both builds.
 */
#ifndef PIGEON_MPC_H
#define PIGEON_MPC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pg_handle pg_handle;
typedef void pg_real_dev;       /* device array of double (libpigeon_hip.so) or float (libpigeon_hip_f32.so) */
/* 64 for libpigeon_hip.so, 32 for libpigeon_hip_f32.so */
int pg_precision_bits(void);

enum pg_status {
    PG_OK = 0,
    PG_ERR_NO_DEVICE = -1,      /* HIP runtime found no usable gfx950 device: the product path NEVER falls back to a CPU */
    PG_ERR_INVALID = -2,        /* bad argument (null pointer, B > capacity, L < 2, ...) */
    PG_ERR_HIP = -3,            /* a HIP call failed; pg_last_error() has the text */
    PG_ERR_STATE = -4           /* call order violated (e.g. no trajectory installed) */
};

/* per-instance solver status words written by pg_solve (the reference never inspects OSQP's status,
 * src/ros_integration.jl:127 "TODO"; the NaN fallback of :134-147 is reproduced by the caller from these) */
enum pg_solve_status {
    PG_SOLVED = 1,
    PG_MAX_ITER = 2,            /* iteration cap reached before the tolerances were met */
    PG_NUMERICAL = 3,           /* NaN/Inf met in the data or the iterates (e.g. the H4 hazard: other-car speed 0) */
    PG_INFEASIBLE_X0 = 4,       /* a hard bound on the fixed first node is violated (Ux_1 outside [V_min,V_max], Fx_1 < Fx_min) */
    PG_SOLVED_UNVERIFIED = 5    /* (polish on) the interior point met its tolerances but NO active-set round verified a KKT point (pg_get_polish_info = -1): the control is
                                 * the interior-point iterate at its rounding floor -- within 1e-6 of the optimum on every tracking batch tested, up to 2e-3 off in the wide
                                 * random regime of tests/test_gpu_fuzz.py.  A caller that treats it like PG_SOLVED gets round-2 behaviour; with polish = 0 the interior
                                 * point's own convergence is all there is and the status stays PG_SOLVED */
};
/* "the solver returned an answer that met its tolerances" (verified or not): what a caller that used to test status == PG_SOLVED most likely means, e.g. the `solved`
 * decision of the ROS loop (src/ros_integration.jl:127-147).  With the polish on, ~1 % of the N = 50 lateral benchmark instances end PG_SOLVED_UNVERIFIED.
 * (The warm start of the next step accepts both in k_solve_lat; k_solve, the coupled solver, starts an unverified instance cold: its multipliers at hand-over
 * belong to a working set that did not verify.) */
#define PG_IS_SOLVED(st) ((st) == PG_SOLVED || (st) == PG_SOLVED_UNVERIFIED)

/* vehicle dictionary: src/vehicles.jl:1-59; field sets of src/vehicle_dynamics.jl:7-29,272-292 */
typedef struct pg_vehicle {
    double G, m, Izz, L, a, b, h, mu, Caf, Car, Cd0, Cd1, Cd2;
    double fwd_frac, rwd_frac, fwb_frac, rwb_frac;
    double Fx_max, Fx_min, Px_max, delta_max, kappa_max;
} pg_vehicle;

/* CoupledControlParams: src/coupled_lat_long.jl:1-40 */
typedef struct pg_control_params {
    double V_min, V_max, k_V, k_s, deltadot_max;
    double Q_ds, Q_dpsi, Q_e, W_beta, W_r, W_HJI;
    double R_delta, R_ddelta, R_Fx, R_dFx;
    int32_t N_HJI;
    int32_t _pad;
} pg_control_params;

/* keyword arguments of CoupledTrajectoryTrackingMPC (src/coupled_lat_long.jl:42-43) + build-defined knobs */
typedef struct pg_config {
    pg_vehicle vehicle;
    pg_control_params control;
    int32_t N_short, N_long;        /* default 10, 20 */
    double dt_short, dt_long;       /* default 0.01, 0.2 */
    int32_t use_correction_step;    /* default 1 */
    int32_t rk4_substeps;           /* sub-steps of the RK4 `propagate` inside linearize (third-party default 10) */
    double hji_eps;                 /* HJI_eps, src/model_predictive_control.jl:67 (0.05) */
    int32_t batch_capacity;         /* maximum B */
    int32_t device;                 /* HIP device ordinal */
    int32_t ipm_max_iter;           /* interior-point iteration cap (default 40) */
    int32_t formulation;            /* PG_COUPLED (src/coupled_lat_long.jl) or PG_DECOUPLED (src/decoupled_lat_long.jl) */
    double ipm_tol;                 /* complementarity / infeasibility tolerance of the interior point (default 1e-12 in the fp64 library, 1e-5 in the fp32 one) */
    double ipm_mu0;                 /* initial barrier parameter (default 100) */
    int32_t walls;                  /* BUILD-DEFINED EXTENSION (BASELINE config 5 "both_walls"; the reference snapshot carries edge_L/edge_R through
                                     * TrajectoryTube, src/trajectories.jl:19-20,33, but no constraint reads them, README.md:54): 1 adds to the DECOUPLED
                                     * formulation, at nodes t = 2..N+1, the rows  e_t <= edge_L(s_t) + sw_t,  e_t >= edge_R(s_t) - sw_t,  sw_t >= 0  and the
                                     * cost  wall_weight * dt_t * sw_t  (soft, like the stability-envelope rows of :193-211); 0 (default) = the reference's QP */
    int32_t allow_f32_long_lateral; /* (occupies what was explicit padding up to round 5: the layout is unchanged, pg_default_config* zero it)  The fp32 library REFUSES the decoupled
                                     * formulation with more than 32 intervals at pg_create unless this is 1: single precision on the open-loop unstable 8 s lateral horizon solves
                                     * 99.7 % of the N = 50 benchmark batch with the applied steering up to 6e-3 rad off the exact optimum -- fp64 is the precision
                                     * for that configuration, and a caller has to ask for the other one explicitly.  Ignored by the fp64 library */
    double wall_weight;             /* linear penalty on the wall slack per second (default 1000) */
    int32_t polish;                 /* 1 (default, both formulations): after the interior point has converged, an active-set polish (OSQP's `polish`, off in the reference's settings,
                                     * src/coupled_lat_long.jl:201-203) solves the equality-constrained problem on the detected active set with the same Riccati passes
                                     * and verifies primal/dual feasibility; removes the sqrt(mu) error of nearly degenerate rows.  0 = interior-point iterate as is */
    int32_t _pad3;
    double polish_rho;              /* penalty on the active rows inside the polish solves (default 1e7 in the fp64 library, 1e3 in the fp32 one; k_solve_lat, the lateral formulation's kernel for
                                     * horizons beyond 20 intervals, multiplies it by 1e3 in fp64: see pg_solve_lat.hip) */
    double polish_tol;              /* feasibility tolerance of the polish verification (default 1e-9 / 1e-4) */
    double polish_ipm_tol;          /* with polish = 1 the interior point first stops at this (looser) tolerance and hands over to the polish (default 3e-6 / 1e-4);
                                     * if the polish cannot verify an active set from there (or the sets cycle), the interior point resumes down to ipm_tol and the polish
                                     * gets a second and last chance; if that fails too the interior-point iterate is the answer, exactly as with polish = 0.
                                     * Values <= ipm_tol disable the early hand-over */
    int32_t warm_polish;            /* 1 (default): an instance whose previous step ended in a solved QP (solved flag set, status PG_SOLVED -- the lateral kernel also accepts PG_SOLVED_UNVERIFIED) first tries the polish from that
                                     * step's active set and multipliers on the new QP data -- the counterpart of the reference's OSQP warm start
                                     * (src/coupled_lat_long.jl:218).  A verified round is the exact optimum of the new QP (iters = 0 then); otherwise the interior point runs
                                     * as for a cold instance.  Ignored when polish = 0 */
    int32_t cold_guess;             /* rounds (default 8; 0 = off) a COLD instance may spend on the polish started from the EMPTY active set before the interior point is
                                     * called: round 1 is the unconstrained LQ optimum, violated rows join the set, rows with negative multipliers leave it.  A verified round
                                     * is the exact optimum (iters = 0 then), as for the warm start; most instances of a tracking problem have a handful of active rows and
                                     * verify within 1-4 rounds (a rate-limited steering ramp: up to ~10), each the price of one interior-point iteration.  Instances whose rounds do not verify (or cycle) run the
                                     * interior point exactly as with cold_guess = 0.  Ignored when polish = 0 */
} pg_config;

enum pg_formulation { PG_COUPLED = 0, PG_DECOUPLED = 1 };

/* Layout check for hand-written mirrors of this header (ctypes, Julia): fills out[0..] with sizeof(pg_config), sizeof(pg_vehicle), sizeof(pg_control_params)
 * and the byte offsets inside pg_config of: control, N_short, dt_short, use_correction_step, hji_eps, batch_capacity, ipm_max_iter, formulation, ipm_tol, ipm_mu0,
 * walls, wall_weight, polish, polish_rho, polish_tol, polish_ipm_tol, warm_polish, cold_guess; then offsetof(pg_control_params, N_HJI) and offsetof(pg_vehicle, kappa_max).
 * Returns the number of entries (23); out may be NULL, at most n entries are written. */
int pg_abi_layout(int32_t* out, int32_t n);

/* X1() and the default keyword values of the reference constructors (coupled formulation) */
int pg_default_config(pg_config* cfg);
/* DecoupledTrajectoryTrackingMPC(vehicle, trajectory; ...) defaults: src/decoupled_lat_long.jl:18-33.  The lateral formulation uses the
 * fields V_min, V_max, k_V, k_s, deltadot_max, Q_dpsi, Q_e, W_beta, W_r, R_delta, R_ddelta of pg_control_params; delta is NOT normalised
 * (pg_get_u_normalization returns (1,1)); pg_get_next_control returns delta from the QP and Fx from the seeded node 2 (:275-278).
 * The warm branch of the NODES does not exist in this formulation (:52-104 always re-seeds); the solver warm-starts (WarmStart = true, :139: pg_config.warm_polish).
 * The HJI row is not part of it. */
int pg_default_config_decoupled(pg_config* cfg);

/* CoupledTrajectoryTrackingMPC(vehicle, trajectory; ...)  src/coupled_lat_long.jl:42-60 (trajectory installed separately) */
int pg_create(const pg_config* cfg, pg_handle** out);
int pg_destroy(pg_handle* h);
const char* pg_last_error(const pg_handle* h);
int pg_get_config(const pg_handle* h, pg_config* out);
/* u_normalization, src/coupled_lat_long.jl:199 */
int pg_get_u_normalization(const pg_handle* h, double out[2]);

/* mpc.trajectory = TrajectoryTube(t,s,V,A,E,N,psi,kappa,theta,phi,edge_L,edge_R)  src/trajectories.jl:8-44; src/ros_integration.jl:53 */
int pg_set_trajectory(pg_handle* h, int32_t L, const double* t, const double* s, const double* V, const double* A, const double* E,
                      const double* N, const double* psi, const double* kappa, const double* theta, const double* phi,
                      const double* edge_L, const double* edge_R);
/* Batched form of the same assignment when instances track DIFFERENT references (one controller per (x0, reference trajectory) pair,
 * src/model_predictive_control.jl:36 `trajectory::TrajectoryTube{T}` is a per-controller field): install a library of n_traj tubes once,
 * then select one per instance.  channels is [n_traj][10][Lmax] doubles, channel order t, s, V, A, E, N, psi, kappa, edge_L, edge_R
 * (src/trajectories.jl:8-21); L[k] <= Lmax is the node count of trajectory k (the tail of a shorter one is ignored).
 * pg_set_trajectory(...) == a library of one; installing a library drops any previous index. */
int pg_set_trajectories(pg_handle* h, int32_t n_traj, int32_t Lmax, const int32_t* L, const double* channels);
/* index[b] in [0, n_traj): the trajectory instance b tracks.  Persistent across steps until the library or the index is replaced.
 * With n_traj > 1 the phase calls return PG_ERR_STATE until an index covering the current batch is installed. */
int pg_set_trajectory_index(pg_handle* h, int32_t B, const int32_t* index);

/* mpc.HJI_cache = HJICache(grid_knots, V_raw, gradV_raw)  src/HJI_computation.jl:26-57.  V is column-major (dim 1 fastest),
 * gradV is 7 floats per node in the same node order.  Without a grid the safety row is inactive (M = 0, b = 1). */
int pg_set_hji_grid(pg_handle* h, const int32_t dims[7], const float* knots_concat, const float* V, const float* gradV);
int pg_clear_hji_grid(pg_handle* h);

/* mpc.solved = false (src/ros_integration.jl:34,41,147): mask[b] != 0 resets instance b; mask == NULL resets all */
int pg_reset(pg_handle* h, const uint8_t* mask);

/* mpc.current_state / current_control / other_car_state / time_offset (src/ros_integration.jl:50-53,76-78,155):
 * state [B][6] (E,N,psi,Ux,Uy,r), control [B][3] (delta,Fxf,Fxr), t0 [B], other_car [B][4] (E,N,psi,V) or NULL (zeros),
 * time_offset [B] (NaN = path-tracking mode) or NULL (all NaN).  Copies host -> device. */
int pg_set_inputs(pg_handle* h, int32_t B, const double* state, const double* control, const double* t0, const double* other_car,
                  const double* time_offset);
/* same, inputs already resident in device memory (HBM) */
int pg_set_inputs_dev(pg_handle* h, int32_t B, const pg_real_dev* state_dev, const pg_real_dev* control_dev, const double* t0_dev,
                      const pg_real_dev* other_car_dev, const double* time_offset_dev);

/* the five reference calls, each over the whole batch, device-resident intermediates */
int pg_compute_time_steps(pg_handle* h);              /* compute_time_steps!           model_predictive_control.jl:17-30 */
int pg_compute_linearization_nodes(pg_handle* h);     /* compute_linearization_nodes!  coupled_lat_long.jl:62-142 */
int pg_update_qp(pg_handle* h);                       /* update_QP!                    coupled_lat_long.jl:315-368 (+ HJI_computation.jl:160-170) */
int pg_solve(pg_handle* h);                           /* solve!                        model_predictive_control.jl:76 */
int pg_get_next_control(pg_handle* h, double* u_out); /* get_next_control              coupled_lat_long.jl:370-374; u_out [B][3] (delta,Fxf,Fxr), host */
int pg_get_next_control_dev(pg_handle* h, pg_real_dev* u_out_dev);
/* The control the ROS loop actually sends (src/ros_integration.jl:114-124): when the instance is in trajectory mode (time_offset not NaN),
 * use_hji_policy is set and the looked-up value V <= HJI_eps, the HJI fallback policy optimal_control(...) (src/HJI_computation.jl:133-158:
 * bang-bang steer + 50-point Fx line search) replaces get_next_control(mpc).  Call after pg_update_qp/pg_solve (it reuses that step's lookup).
 * u_out [B][3] (delta,Fxf,Fxr); source [B] (may be NULL): 0 = MPC control, 1 = HJI policy, 2 = V <= eps but the policy is switched off
 * ("with a feather", :120-123); u2_policy [B][2] (may be NULL) = (delta_opt, Fx_opt) of optimal_control regardless of the selection.
 * Without a grid V = +Inf and the MPC control is returned.  Coupled formulation only. */
int pg_get_next_control_hji(pg_handle* h, int32_t use_hji_policy, double* u_out, int32_t* source, double* u2_policy);
int pg_get_next_control_hji_dev(pg_handle* h, int32_t use_hji_policy, pg_real_dev* u_out_dev, int32_t* source_dev);

/* all five for every instance: host buffers in, host buffers out (status/iters may be NULL).
 * A batch that fills the handle (B == batch_capacity) travels in one copy per direction.  Opt-in (pg_set_option "graph" = 1): when such a batch has at most 256
 * instances and every one of them is warm, the whole step -- copy in, the kernels of a warm step, copy out -- is captured once into a hipGraph and replayed
 * (re-captured whenever something its launches depend on has changed; results identical to the ordinary launches; pg_get_phase_ms has no timing for replayed
 * steps; with no stream installed the graph runs on a blocking stream of its own, ordered against the null stream).  Measured: 2-7 % per step, against ~7 ms for
 * every capture -- worth it for a long run on one path, not for a loop that re-installs its path every few steps; hence off by default. */
int pg_step(pg_handle* h, int32_t B, const double* state, const double* control, const double* t0, const double* other_car,
            const double* time_offset, double* u_out, int32_t* status, int32_t* iters);
/* the four compute phases + control extraction on the inputs last installed; nothing crosses PCIe.  u_out_dev may be NULL. */
int pg_step_dev(pg_handle* h, pg_real_dev* u_out_dev);

/* simulate(mpc, q0, u0, dt)  src/model_predictive_control.jl:80-100 for every instance, entirely on the device (no host round trip between
 * steps): per step  record -> the four compute calls -> state = propagate(dynamics, state, StepControl(dt, old control)) -> control = get_next_control
 * -> t = the next element of the loop's range (see pg_simulate_clock below).  Starts from the inputs last installed (pg_set_inputs*: state, control, t0, time_offset) and leaves the final ones there
 * (pg_get_state reads them).  state_hist_dev [steps][B][6] / control_hist_dev [steps][B][3] may be NULL.  Asynchronous on the handle's stream. */
int pg_simulate_dev(pg_handle* h, int32_t steps, double dt, pg_real_dev* state_hist_dev, pg_real_dev* control_hist_dev);
/* The loop variable of that rollout: `for t in 0:dt:mpc.trajectory.t[end]` (src/model_predictive_control.jl:87) is a Julia RANGE -- element k is ONE rounding of k dt with dt lifted
 * to its exact rational (0.01 = 1/100) when dt and the path's end time have one, `fl(k dt)` otherwise -- not the accumulation t += dt (which is 0.2900000000000001 at step 29 and
 * then puts the long horizon of :23 a whole dt_long later).  pg_simulate_dev forms each instance's time as (t_start .+ (0:dt:t_end))[k + 1] with t_start the time installed by
 * pg_set_inputs* and t_end the last time of the installed trajectory (trajectory 0 of a library); consecutive calls with the same dt continue the clock, pg_set_inputs* restarts it.
 * This entry point returns those times on the host, out[k * B + b] for k < steps, so that a host-driven loop (pg_step per step) can feed the same t0 the device loop uses.
 * Option "time_grid_naive" = 1: t_start + accumulated dt, as in rounds 1-5.  (A restatement of Julia 1.0's Base range arithmetic that could not be executed here.) */
int pg_simulate_clock(pg_handle* h, double dt, int32_t steps, int32_t B, const double* t_start, double* out);
/* current device-resident inputs: state [B][6], control [B][3], t0 [B] (host pointers, any may be NULL) */
int pg_get_state(pg_handle* h, double* state, double* control, double* t0);

/* stream to launch on (hipStream_t as void*); NULL = the null stream.  pg_set_inputs and pg_step are ASYNCHRONOUS on the handle's stream (copies from / into one pinned
 * staging buffer, synchronised at the next entry that needs it): switching streams first waits for whatever is still queued on the old one. */
int pg_set_stream(pg_handle* h, void* hip_stream);
/* Fused step: pg_step / pg_step_dev / pg_simulate_dev can run update_QP! and solve! of the coupled formulation (N <= 32) in ONE kernel -- the wavefront that
 * solves an instance linearises it first (same device functions: results are bit-identical either way; the QP data are still written and pg_get_qp reads them).
 * mode 0 = never (default), 1 = always, 2 = for batches of >= 1024 instances in which every instance is warm (closed loop).
 * Measured on MI355X: +7 % on the cold benchmark batch (skidpadoval), -5..-8 % on the other paths and in closed loop (EXPERIMENTS.md 4.1) -- it pays only where a few
 * slow instances dominate the solve kernel.  The four compute calls invoked one by one are never fused. */
int pg_set_fusion(pg_handle* h, int32_t mode);
/* Pipelined nodes + update_QP (build-defined; no counterpart in the reference): for batches of 2304..8192 instances with cold instances, coupled formulation,
 * pg_step / pg_step_dev / pg_simulate_dev run compute_linearization_nodes! and update_QP! as ONE launch in which the linearisation of
 * interval t starts as soon as nodes t, t + 1 of its instances are seeded (the cold seeding is a serial recurrence over the nodes: 0.18 ms of latency on 6 % of the
 * chip that the linearisation of the early intervals now runs under).  With a safety row installed its (M, b) are computed before that launch.  Same device
 * functions on the same arguments: nodes and QP data are bit-identical in fp64; in the fp32 library the nodes are, the QP data agree to fp32 rounding.
 * mode 1 = on where it applies (default), 0 = never.  The compute calls invoked one by one are never pipelined. */
int pg_set_pipeline(pg_handle* h, int32_t mode);
/* The wavefronts of that launch that linearise an interval WAIT for the wavefronts that seed its nodes.  The wait is bounded (20 ms of wall-clock time); a wavefront
 * that gives up -- the seeding wavefronts not resident before it: a dispatch order the launch does not control, a debugger, a time-sliced or counter-serialised run --
 * raises a device flag, and the launch-per-phase kernels queued behind the pipelined launch (predicated on that flag) redo update_QP! for the batch: such a step is
 * late, not wrong, and nothing is reported to the caller but this cumulative count of wavefronts that gave up (synchronises the handle's stream). */
int pg_get_pipeline_fallbacks(pg_handle* h, int64_t* count);
int pg_synchronize(pg_handle* h);

/* Build-defined options of a handle, by name (no counterpart in the reference; every one has a default under which the library behaves as documented above).  The library
 * reads NOTHING from the process environment: what a handle does depends on its pg_config and on the options set here.  Unknown name, read-only name or a value out of
 * range: PG_ERR_INVALID.  Integer options take integral values.  Options take effect at the next launch (hji_cell_dims: at the next pg_set_hji_grid).
 *   coupled solve kernel (k_solve):
 *     "solve_split" 0/1 (1)      rounds-only kernel + list-mode full kernel instead of one kernel (never with a safety row installed)
 *     "clip_guess" 0/1 (1)       first roll-out of a cold instance clips the steering rate; the clipped transitions are its first working set
 *     "clip_stops" 0/1 (0)       ... and at the steering stops too (measured: fewer rounds, a longer launch; EXPERIMENTS.md 11.3)
 *     "ck_riccati" 0/1 (1)       the matrix recursion of a later round restarts at a checkpoint behind the rows that changed (fp64)
 *     "warm_trivial_cold" 0/1 (1)  a warm instance whose previous working set was empty starts like a cold one
 *     "hji_seed" 0..4 (0), "hji_rounds" 0..64 (0)   seeded working sets for instances whose safety row is violated at the current control (experiment, off)
 *   launch shape:
 *     "pipe_min" (2304), "pipe_max" (8192, at most: 256 nodes wavefronts of 32 instances)  batch sizes the pipelined nodes + update_QP launch serves (pg_set_pipeline)
 *     "pipe_first" 0..64 (0)     short-horizon intervals whose wavefronts go first in that launch (0: as many as fill the SIMDs the recurrence leaves free -- all ten at B = 4096;
 *                                measured: fewer is slower, 0.317 / 0.323 / 0.328 / 0.344 ms for 10 / 8 / 6 / 2)
 *     "pipe_pub_short" (3), "pipe_pub_long" (5)   the recurrence of that launch publishes its progress after every n-th node of the short / long horizon (a publication is a
 *                                device-scope release, ~3 us on the serial chain: every node of the short horizon 0.317 -> 0.345 ms)
 *     "lin_lanes" 1/2 (1)        lanes per (instance, interval) of the large-batch linearisation
 *     "phase_timing" 0/1 (0)     1 = pg_step_dev records the HIP events pg_get_phase_ms reads (four event records per step on the handle's stream: measured 13-25 us per step, 2-4 % of a
 *                                4096-instance step); 0 = no instrumentation, pg_get_phase_ms returns PG_ERR_STATE
 *     "graph" 0/1 (0)            pg_step of a small warm batch as one hipGraph launch (see pg_step)
 *     "time_grid_naive" 0/1 (0)  0 = the time axes as Julia's RANGES give them (src/model_predictive_control.jl:25-26: `t0 .+ dt_short*(0:N_short)`, `t0_long .+ dt_long*(1:N_long)`,
 *                                and :87, `for t in 0:dt:trajectory.t[end]` in pg_simulate_dev): reference value and step in twice the working precision, dt lifted to its exact
 *                                rational (0.01 = 1/100, 0.2 = 1/5), every element ONE rounding -- a restatement of Julia 1.0's Base that could not be executed here;
 *                                1 = `t0 + dt*i` with two roundings and `t += dt` in the closed loop (the form of rounds 1-5: differs by an ulp of ts now and then, which the
 *                                discontinuous `ceil` of :23 can turn into a different lattice point)
 *     "hji_cell_dims" 3/5/7 (3)  corners per cell record of the HJI table = 2^value (256 B / 1 KiB / 4 KiB records)
 *   lateral solve kernel (decoupled formulation):
 *     "lateral_solver" 0/1/2 (0) 0 = k_solve_lat beyond 20 intervals or with the polish off, else the embedding in k_solve; 1 = k_solve_lat; 2 = the embedding
 *     "lat_workspace" 0/1        k_solve_lat keeps its row state in the per-wavefront workspace at every horizon (default: beyond 32 intervals, or with walls beyond 16)
 *     "lat_pin" 0/1 (1)          a held steering-rate row pins the input of its stage exactly in the polish (0: held through the augmented Lagrangian like every other row)
 *     "lat_pack_only" 0/1 (1)    update_QP! of a handle solved by k_solve_lat writes the packed stage records only (the embedded block pg_get_qp returns is built on demand)
 *     "lat_split" 0/1 (1)        a step in which every instance is warm runs as two launches (warm attempts, then the cold solves of what they left)
 *     "lat_handover" 0/1 (1)     STRAGGLER HAND-OVER of a cold launch (batches of >= "lat_hand_batch" (1025) instances whose row state lives in the workspace: N > 16): the launch stops
 *                                after "lat_hand_cap" trips through the solver's loop (0 = 16 with the wall rows, 11 without), files its unfinished instances, and a second launch
 *                                resumes them with ONE instance per wavefront.  Same verified KKT points as the single launch (measured 4e-8 apart at most); 2.96-3.03 -> 2.61 ms on
 *                                the N = 50 + walls batch of 4096.  0 = one launch, as in round 5.  The trip rule depends on the data only: the same call returns the same bits.
 *     "lat_hand_target" (0)      > 0: stop the first launch once at most this many instances of the batch are unfinished instead (counted on the device, not before "lat_hand_min" (8)
 *                                trips).  Adapts to the batch (1500: 2.58 ms on the batch above, vail + walls 2.09 against 2.19) -- but WHEN a wavefront sees the count is a matter of
 *                                timing: answers then differ by ~1e-8 from run to run (two verified KKT points of the same QP, resumed a trip earlier or later)
 *                                ("lat_hand_work" > 0 with "lat_hand_w0": a third, deterministic rule kept for A/B -- stop when w0 + #unfinished instances, summed over the wavefront's
 *                                trips, reaches the budget: measured worse than the trip count at every weight, EXPERIMENTS 12.8)
 *     "lat_single_max" (1024)    cold lateral batches of at most this many instances (horizons beyond 16 intervals) run one instance per WAVEFRONT from the start: a trip through the
 *                                solver's loop costs 57 us instead of 84 (0: never).  The same arrangement serves the list a warm step's attempts leave, when it is that short (the length
 *                                is a device word: both arrangements are queued, one of them returns at once)
 *     "lat_aux_gate" 0/1 (1)     the serial passes write what a pinned row's multiplier is read from only while an instance of the wavefront is in a polish
 *     "nodes_serial" 0/1 (0)     1 = the cold node seeding (both formulations) commits ONE node per pass: the reference's serial recurrence exactly (the default runs 2 / 8 lanes per
 *                                instance ahead on the commanded acceleration and commits the nodes whose solve returned it: identical to 1e-12, 4e-6 in fp32; parity tests use this)
 *     "lat_rho_scale" (1e3 in fp64, 1 in fp32)  penalty of held rows = polish_rho x this;   "lat_mu0_cost" (10), "lat_far_cost" (3e4), "lat_polish2" 0/1 (1),
 *     "lat_polish_rounds" (3), "lat_settle" 0..2 (0), "lat_warm_rounds" (2), "lat_wipm" 0/1 (0), "lat_wmu" (1e-2), "lat_wtau" (1e-4)   see pg_solve_lat.hip
 *   read-only (pg_get_option): "stat_pipelined_launches", "stat_split_solve_launches", "stat_single_solve_launches", "stat_lat_two_launch_solves", "stat_lat_handover_solves", "stat_lat_one_per_wavefront_solves" -- how many launches of
 *     this handle took the path named (tests assert that the path they mean to cover is the one that ran); "stat_whole_batch_solves" -- counted ON THE DEVICE: launches in
 *     which the full k_solve took the whole batch because the previous launch had left instances for the interior point (reading it drains the stream);
 *     "lateral_solver_in_use" (1 = k_solve_lat, 2 = embedding).
 * The diagnostic build (libpigeon_hip_diag.so, -DPG_DIAG; never shipped) adds "diag_pipe_fault" (fault injection for the pipelined launch), "diag_instance",
 * "diag_lin_groups", "diag_timeline". */
int pg_set_option(pg_handle* h, const char* name, double value);
int pg_get_option(pg_handle* h, const char* name, double* value);

/* ---- read-backs for parity tests and logging (host pointers, any may be NULL) ---------------------------------- */
/* ts [B][N+1], dt [B][N], prev_ts [B][N+1] */
int pg_get_time_steps(pg_handle* h, double* ts, double* dt, double* prev_ts);
/* qs [B][N+1][6], us [B][N+1][2] (delta, Fx in physical units), ps [B][N+1][4] (V, kappa, 0, 0) */
int pg_get_nodes(pg_handle* h, double* qs, double* us, double* ps);
/* path_coordinates of the current states: sep [B][3] = (s, e, t)   src/trajectories.jl:71-94 */
int pg_get_path_coordinates(pg_handle* h, double* sep);
/* refreshed QP data of instances [b0, b0+n): n blocks of pg_qp_len() doubles laid out as
 * A[N][6][6] B0[N][6][2] Bf[N][6][2] c[N][6] H[N][4][2] G[N][4] dmin[N] dmax[N] fxmax[N] ddmin[N] ddmax[N] dt[N] q_curr[6] u_curr[2] M_hji[2] b_hji
 * (the numeric content update_QP! writes into the Parametron parameters, coupled_lat_long.jl:323-366; B's scaled by u_normalization) */
int pg_qp_len(const pg_handle* h);
int pg_get_qp(pg_handle* h, int32_t b0, int32_t n, double* out);
/* the inverse: install QP data for instances [b0, b0+n) in the same layout (what setting the Parametron parameters by hand is to the reference); pg_solve then solves
 * exactly these problems.  For replaying recorded QPs and for solver tests on constructed (e.g. degenerate) problems; a step (pg_update_qp, pg_step*) overwrites them. */
int pg_set_qp(pg_handle* h, int32_t b0, int32_t n, const double* in);
/* primal solution: x [B][N+1][8] = (q (6), normalised u (2)) per node; sigma [B][N][3] = (sigma1, sigma2, sigma_HJI of node k+1) */
int pg_get_solution(pg_handle* h, double* x, double* sigma);
/* status [B] (pg_solve_status), iters [B], active [B][N] bit masks over the 16 stage rows (row order in DESIGN.md), mu [B] final gap */
int pg_get_solve_info(pg_handle* h, int32_t* status, int32_t* iters, uint16_t* active, double* mu);
/* outcome of the active-set polish per instance, [B]: 0 = not run (polish off, or the interior point did not converge), k >= 1 = verified in round k (the
 * solution is the exact optimum on its active set), -1 = did not verify (the interior-point iterate at ipm_tol was kept) */
int pg_get_polish_info(pg_handle* h, int32_t* polish);
/* multipliers of the inequality rows as the last solve left them, lam [B][N][16], indexed like the bits of the `active` masks (the next step's warm start reads them).
 * A verified instance (pg_get_polish_info >= 1): the multipliers of its verified working set, 0 off the set -- with the masks, the dual half of the KKT point; the
 * CANONICAL active-set rule of the parity tests is "bit set AND multiplier > 1e-6" (a row held at its bound with a zero multiplier is degenerate: the QP does not say
 * which side of "active" it is on), the same rule the oracle applies to its own multipliers.  An unverified instance: the interior point's multipliers at the hand-over
 * (k_solve_lat) or the estimates of the last working set tried (k_solve) -- not a certificate. */
int pg_get_multipliers(pg_handle* h, double* lam);
/* milliseconds of the last pg_step_dev per phase: time_steps+nodes, update_qp (linearize, limits, HJI), solve (+extract); HIP events, recorded when the option
 * "phase_timing" is 1 (off by default: PG_ERR_STATE) */
int pg_get_phase_ms(pg_handle* h, float out3[3]);

/* cache[x] for a batch of relative states: HJI_computation.jl:66-72.  x7 [B][7] host; V [B], gradV [B][7] host.  Out of bounds => V=+Inf, gradV=0 */
int pg_hji_lookup(pg_handle* h, int32_t B, const double* x7, double* V, double* gradV);
int pg_hji_lookup_dev(pg_handle* h, int32_t B, const pg_real_dev* x7_dev, pg_real_dev* V_dev, pg_real_dev* gradV_dev);
/* packed form, no temporaries, asynchronous on the handle's stream: out8 [B][8] = (V, gradV[0..6]) per lookup (the kernel's native output) */
int pg_hji_lookup8_dev(pg_handle* h, int32_t B, const pg_real_dev* x7_dev, pg_real_dev* out8_dev);
/* dims of the installed grid (HJICache.grid_knots lengths) */
int pg_hji_grid_dims(pg_handle* h, int32_t dims[7]);
/* 2-D value slices for the RViz consumers, batched: src/rviz.jl:23-40 (update_HJI_values_marker!) and :60-69 (update_HJI_contour_marker!) evaluate
 * cache[HJIRelativeState(x, y, q[3..7])].V at every knot pair (x, y) of grid dimensions 1 and 2.  q7 [B][7] relative states (components 0, 1 are replaced by
 * the knots); V_out [B][n1][n2] = V(X[i], Y[j]);  rgb_out [B][n1][n2][3] or NULL = value_to_RGB(V) (rviz.jl:41-44);  zero-level crossings = the vertex set of
 * contour(X, Y, V, 0) (:63): cross_x [B][n1-1][n2] = x where V changes sign between (X[i], Y[j]) and (X[i+1], Y[j]), NaN where it does not;
 * cross_y [B][n1][n2-1] likewise along y (either may be NULL).  An edge carries a vertex iff exactly one end has V > 0 (Contour.jl's marching-squares rule). */
int pg_hji_slice(pg_handle* h, int32_t B, const double* q7, double* V_out, double* rgb_out, double* cross_x, double* cross_y);
/* compute_reachability_constraint for the installed inputs: M [B][2] (already multiplied by u_normalization), b [B], V [B] */
int pg_get_hji_constraint(pg_handle* h, double* M, double* b, double* V);
/* wall extension (pg_config.walls): (edge_L, edge_R) at nodes 2..N+1 of every instance, [B][N][2]; PG_ERR_STATE when walls are off.
 * The wall slack sw is returned in the third column of pg_get_solution's sigma [B][N][3]. */
int pg_get_walls(pg_handle* h, double* edges);

#ifdef __cplusplus
}
#endif
#endif
