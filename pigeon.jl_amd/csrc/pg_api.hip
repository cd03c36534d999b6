// C ABI of libpigeon_hip.so (declared in include/pigeon_mpc.h).  Host-side handle, device buffers, kernel launches.
// The product path has NO CPU fallback: without a HIP device pg_create fails with PG_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include "pg_kernels.hip"

using namespace pg;

struct pg_handle {
    pg_config cfg;
    DevCfg dc;
    int B = 0;                       // current batch
    int warm_B = 0;                  // instances [0, warm_B) are known to carry a previous solution (solved = true): set by pg_solve, cleared by pg_reset
    hipStream_t stream = nullptr;
    int pipeline = 1;                                         // pg_set_pipeline: 1 = nodes + update_QP of a large batch with cold instances as one pipelined launch (k_nodes_linearize), 0 = never
    bool pipe_fault = false;                                  // -DPG_DIAG builds only (option "diag_pipe_fault"): fault injection for the pipelined launch (its nodes blocks never publish)
    int lin_groups = 0;                                       // -DPG_DIAG builds only (option "diag_lin_groups"): force the lane arrangement of launch_linearize (1, 2, 4, 8 lanes per interval)
    int debug_timeline = 0;                                   // -DPG_DIAG builds only (option "diag_timeline"): pg_debug_solve_cycles records the timeline of the product's kernel
    int lateral_solver = 0;                                   // option "lateral_solver": 0 = by horizon (k_solve_lat beyond 20 intervals or with the polish off), 1 = k_solve_lat, 2 = the embedding in k_solve
    int hji_cell_dims = 3;                                    // option "hji_cell_dims": corners per cell record of the next pg_set_hji_grid = 2^3 (256 B, default), 2^5, 2^7
    int64_t stat_pipelined = 0, stat_split = 0, stat_single = 0, stat_lat_two = 0;      // read-only options "stat_*": launches that took the pipelined nodes + update_QP path / the split solve / the single solve kernel / the two-launch lateral solve
    int solve_parity = 0;                                     // which of the two to-do counters the next solve launch counts into (the other one holds the previous launch's count: see launch_solve)
    bool cnt_cleared = false;                                 // this step's projection kernel has zeroed the counter launch_solve is about to use
    bool lin_done = false;                                    // this step's launch_nodes already linearised (update_and_solve skips update_QP)
    int* d_progress = nullptr;                                // [cap / NODES_IPB + 8] nodes completed per nodes wavefront (k_nodes_linearize), then the fault word of the launch; last entry: fall-backs so far
    int64_t fallback_total = 0; int fallback_seen = 0;        // pg_get_pipeline_fallbacks: 64-bit total kept on the host, last value of the device's 32-bit word
    int fuse = 0;                                             // pg_step_dev / pg_simulate_dev: linearisation fused into the solve kernel (pg_set_fusion): 0 never (default), 1 always, 2 for all-warm batches
    std::string err;
    // device buffers
    real *d_traj = nullptr; int traj_L = 0; int *d_traj_len = nullptr, *d_traj_idx = nullptr; int traj_idx_B = 0;
    real *d_state = nullptr, *d_control = nullptr, *d_other = nullptr;
    double *d_t0 = nullptr, *d_toff = nullptr;            // absolute time stays fp64 in both builds (tdouble)
    // clock of pg_simulate_dev (model_predictive_control.jl:87, `for t in 0:dt:trajectory.t[end]`): start time per instance, the range, and the index of the element t0 holds.
    // pg_set_inputs* restarts it (sim_idx = 0); consecutive pg_simulate_dev calls with the same dt continue it, so 4 + 40 steps see the times of 44
    double* d_tstart = nullptr; JlRange sim_clk{}; int sim_idx = 0; double sim_dt = 0.0, sim_tend = 0.0; double traj_t_end = 0.0;
    int* d_solved = nullptr; uint8_t* d_mask = nullptr;     // d_mask: staging of pg_reset's per-instance mask
    double *d_ts = nullptr, *d_dt = nullptr, *d_prev_ts = nullptr;
    real *d_sep = nullptr, *d_nodes = nullptr, *d_qp = nullptr, *d_naux = nullptr;      // d_naux [cap][NN][4]: arguments of the angles k_nodes leaves to k_nodes_angles
    real *d_x7 = nullptr, *d_vg8 = nullptr, *d_Mb = nullptr;
    real *d_solx = nullptr, *d_sigma = nullptr, *d_u = nullptr, *d_mu = nullptr;
    real* d_lam = nullptr;                                      // [cap][N][16] multipliers of the last solve (warm start of the polish)
    int* d_wfail = nullptr;                                     // [cap] back-off of k_solve_lat's warm attempts (the second half of the d_solved allocation: pg_reset clears both with one fill)
    int pipe_pub_short = 3, pipe_pub_long = 5;                        // options: the recurrence of the pipelined launch publishes every n-th node of the short / long horizon
    int pipe_first = 0;                                               // option "pipe_first": short-horizon intervals that go first in the pipelined launch (0 = as many as fill the SIMDs the recurrence leaves free)
    int pipe_min = 2304, pipe_max = 256 * NODES_IPB;                    // batch sizes the pipelined launch serves (options "pipe_min" / "pipe_max"; its nodes blocks must be resident at once: <= 16384)
    int lin_lpi = 1;                                          // lanes per (instance, interval) of the large-batch linearisation (k_linearize_split / k_nodes_linearize): one lane with all eight
                                                              // directions (option "lin_lanes" = 2: the lane pair of rounds 1-3, for A/B runs; same bits in fp64, rounding-level differences in fp32)
    int* d_todo = nullptr; int split_solve = 1, split_lat = 1;      // d_todo [cap + 8]: instances the rounds-only k_solve leaves to the full kernel; behind them the control words of the solve launches:
                                                                    // [0], [1] two to-do counters used alternately (this launch's count / the previous launch's)
    int phase_timing = 0;                                           // option "phase_timing": 1 = pg_step_dev records the four HIP events pg_get_phase_ms reads (13-25 us of stream time per step at B = 4096: 2-4 %); 0 (default) = no instrumentation on the stream
    real* u_direct = nullptr; bool u_written = false;               // pg_step_dev: the caller's control array for k_solve to write (SolveOut::u_out2), and whether the launch did
    int* d_order = nullptr; int order_B = 0;  // [cap] + 2 counters: launch order filed by the nodes kernels of the current step (likely slow instances first); order_B = batch it is valid for
    real *d_pol_u2 = nullptr, *d_pol_u = nullptr; int* d_pol_src = nullptr;   // HJI fallback policy (HJI_computation.jl:133-158)
    int *d_status = nullptr, *d_iters = nullptr, *d_polish = nullptr; uint16_t* d_active = nullptr;
    // HJI grid
    HjiView hv; float *d_knots = nullptr, *d_hnodes = nullptr, *d_hcells = nullptr; bool has_hji = false;
    hipEvent_t ev[4]; bool ev_ok = false; float phase_ms[3] = {0, 0, 0}; bool timing_valid = false;
    size_t solve_lds = 0; bool solve_ring = false;
    real* d_walls = nullptr;                                    // [cap][N][2] wall extension
    char* d_lat_ws = nullptr; bool lat_mem = false, lat_mem_forced = false;            // k_solve_lat's workspace for horizons beyond 32 intervals (option "lat_workspace" = 1: at every horizon)
    real* d_lat_spc = nullptr;                                  // k_solve_lat's lane-contiguous stage constants (lat_spc_bytes)
    real* d_hand_r = nullptr; int* d_hand_i = nullptr;          // [cap][8] / [cap][16] hand-over records of k_solve_lat's unfinished instances (round 6: SolveOut::hand_mode)
    int lat_single_max = 1024;                                  // option "lat_single_max": cold lateral batches up to this size run ONE instance per wavefront from the start (k_solve_lat<.., 64, 0>)
    int lat_handover = 1, lat_hand_target = 0, lat_hand_min = 8, lat_hand_cap = 0, lat_hand_batch = 1025, lat_hand_work = 0, lat_hand_w0 = 3;      // options "lat_handover" (0 off, 1 = on), "lat_hand_target", "lat_hand_min", "lat_hand_cap", "lat_hand_batch" (smallest batch that hands over)
    int64_t stat_lat_single = 0;                                // read-only option "stat_lat_one_per_wavefront_solves"
    int64_t stat_lat_hand = 0;                                  // read-only option "stat_lat_handover_solves"
    real* d_lat_aux = nullptr;                                  // [cap][64][8] F, Bbar'P Bbar, Bbar'y per stage: what k_solve_lat reads the multiplier of a pinned rate row from
    real* d_lat = nullptr;                                      // [cap][N][LATP] packed stage records of the lateral formulation (k_qp_dec -> k_solve_lat)
    // hipGraph of a whole host-to-host warm step (pg_step of a small batch is launch-bound: one copy in, four kernels, one copy out; captured once, replayed while
    // nothing that the launches depend on has changed -- `sig` is compared field by field before every replay)
    struct StepGraph { hipGraph_t g = nullptr; hipGraphExec_t x = nullptr; hipStream_t own = nullptr; bool disabled = false; bool capturing = false;
                       DevCfg dc; HjiView hv; int B = 0, fuse = 0, pipeline = 0, has_hji = 0, traj_L = 0; hipStream_t user = nullptr; } sg;
    int graph_mode = 0;                                       // option "graph" = 1 turns it on.  OFF by default: capturing costs ~7 ms once and again whenever a launch parameter changes (a re-installed
                                                              // path: every `path` message of the ROS loop), a jitter that a 100 Hz loop minds more than the 4-20 us per step the replay saves
    char* d_in = nullptr; char* d_out = nullptr;             // the five input arrays / (u, status, iters) as ONE allocation each: a batch that fills the handle travels in one copy per direction
    size_t in_bytes = 0, out_bytes = 0, in_dbl_off = 0;        // (layout by capacity: [state 6][control 3][other 4] real, then at in_dbl_off [t0][time_offset] double; [u 3] real, [status][iters] int)
    char* h_stage = nullptr; size_t stage_bytes = 0;            // pinned host staging of pg_set_inputs / pg_step (one stream synchronisation per call instead of one per array)
    real* d_ws4 = nullptr; bool solve_quad = false; size_t solve4_lds = 0;   // k_solve4 (four instances per wavefront)
    bool qp_embedded = true; int lat_pack_only = 1;            // lateral formulation: the embedded QP block is current (k_qp_dec wrote it) / option "lat_pack_only": steps write the packed records only
    bool qp_stale = false;                                    // the lateral solver was switched (option "lateral_solver") after the last update_QP!: pg_solve returns PG_ERR_STATE until the QP data are rebuilt
    bool solve_lat = false; size_t lat_lds = 0;               // lateral formulation: its own kernel k_solve_lat (option "lateral_solver" = 2 keeps the embedding in k_solve)
};

#define HIPCHK(h, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { (h)->err = std::string(#call) + ": " + hipGetErrorString(e_); return PG_ERR_HIP; } } while (0)
#define LAUNCH_CHECK(h) do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) { (h)->err = std::string("kernel launch: ") + hipGetErrorString(e_); return PG_ERR_HIP; } } while (0)
#define REQUIRE(h, cond, msg) do { if (!(cond)) { if (h) (h)->err = (msg); return PG_ERR_INVALID; } } while (0)

static std::string g_create_error;

// Host <-> device transfers of path data.  The ABI speaks double on the host side; device buffers are `real` (double in libpigeon_hip.so,
// float in libpigeon_hip_f32.so).  *_dev entry points take device pointers in the library's own element type (times are always double).
static int up(pg_handle* h, real* dst, const double* src, size_t n) {            // returns after the host buffer has been consumed
#ifdef PG_F32
    std::vector<float> tmp(n);
    for (size_t i = 0; i < n; i++) tmp[i] = (float)src[i];
    HIPCHK(h, hipMemcpyAsync(dst, tmp.data(), n * sizeof(float), hipMemcpyHostToDevice, h->stream));
#else
    HIPCHK(h, hipMemcpyAsync(dst, src, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
#endif
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return PG_OK;
}
static int down(pg_handle* h, double* dst, const real* src, size_t n) {          // returns after the copy has completed
    if (!dst) return PG_OK;
#ifdef PG_F32
    std::vector<float> tmp(n);
    HIPCHK(h, hipMemcpyAsync(tmp.data(), src, n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (size_t i = 0; i < n; i++) dst[i] = (double)tmp[i];
#else
    HIPCHK(h, hipMemcpyAsync(dst, src, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
#endif
    return PG_OK;
}
static int down_raw(pg_handle* h, void* dst, const void* src, size_t bytes) {      // integer / time buffers: same type on both sides
    if (!dst) return PG_OK;
    HIPCHK(h, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return PG_OK;
}
static void fill_dev_params(DevCfg& C, const pg_config* cfg) {
    const pg_vehicle& v = cfg->vehicle; DevVehicle& V = C.veh;
    V.G = (real)v.G; V.m = (real)v.m; V.Izz = (real)v.Izz; V.L = (real)v.L; V.a = (real)v.a; V.b = (real)v.b; V.h = (real)v.h; V.mu = (real)v.mu;
    V.Caf = (real)v.Caf; V.Car = (real)v.Car; V.Cd0 = (real)v.Cd0; V.Cd1 = (real)v.Cd1; V.Cd2 = (real)v.Cd2;
    V.fwd_frac = (real)v.fwd_frac; V.rwd_frac = (real)v.rwd_frac; V.fwb_frac = (real)v.fwb_frac; V.rwb_frac = (real)v.rwb_frac;
    V.Fx_max = (real)v.Fx_max; V.Fx_min = (real)v.Fx_min; V.Px_max = (real)v.Px_max; V.delta_max = (real)v.delta_max; V.kappa_max = (real)v.kappa_max;
    const pg_control_params& u = cfg->control; DevControl& U = C.cp;
    U.V_min = (real)u.V_min; U.V_max = (real)u.V_max; U.k_V = (real)u.k_V; U.k_s = (real)u.k_s; U.deltadot_max = (real)u.deltadot_max;
    U.Q_ds = (real)u.Q_ds; U.Q_dpsi = (real)u.Q_dpsi; U.Q_e = (real)u.Q_e; U.W_beta = (real)u.W_beta; U.W_r = (real)u.W_r; U.W_HJI = (real)u.W_HJI;
    U.R_delta = (real)u.R_delta; U.R_ddelta = (real)u.R_ddelta; U.R_Fx = (real)u.R_Fx; U.R_dFx = (real)u.R_dFx; U.N_HJI = u.N_HJI;
}

// Lateral formulation: which kernel solves it, and the buffers that kernel needs.  Called by pg_create and by pg_set_option("lateral_solver" / "lat_workspace").
static int configure_lateral(pg_handle* h, std::string* why) {
    DevCfg& C = h->dc; const pg_config* cfg = &h->cfg; const int N = C.N; const size_t cap = (size_t)cfg->batch_capacity;
    const bool want = h->lateral_solver == 1 || (h->lateral_solver != 2 && !(cfg->polish && N <= 20));
    h->solve_lat = cfg->formulation == PG_DECOUPLED && want && !h->solve_quad;
    if (!h->solve_lat) { C.lat_pack = nullptr; return PG_OK; }
    if (!h->d_lat && hipMalloc((void**)&h->d_lat, cap * N * LATP * sizeof(real)) != hipSuccess) { *why = "hipMalloc failed for the packed lateral stage records"; return PG_ERR_HIP; }
    C.lat_pack = h->d_lat;
    if (!h->d_lat_aux && hipMalloc((void**)&h->d_lat_aux, ((cap + 1) * 64 * LAT_AUX + 8) * sizeof(real)) != hipSuccess)      /* (+ 1: the spare block idle lane groups of a resumed launch write) */ { *why = "hipMalloc failed for k_solve_lat's multiplier block"; return PG_ERR_HIP; }
    C.lat_aux = h->d_lat_aux;
    C.lat_zero = h->d_lat_aux + (cap + 1) * 64 * LAT_AUX;      // (+ 8 stored zeros: DevCfg::lat_zero)
    if (hipMemset((void*)C.lat_zero, 0, 8 * sizeof(real)) != hipSuccess) { *why = "hipMemset failed"; return PG_ERR_HIP; }
    // (round 4: with the wall rows the two-slot register variant spills 720 B per lane since the warm start was added -- 1.57 ms at N = 30 against 1.36 ms through the
    // workspace; without them the registers still win, 0.94 against 1.01 ms)
    // (round 6: horizons of 17..32 intervals WITHOUT the wall rows used the two-slot register instantiation, k_solve_lat<2, .., false> -- 512 registers + 300 B of scratch, the
    // largest code of the kernel.  With this round's edits hipcc 7.2 allocated one accumulation register to two live values there (rocgdb, precise memory violations on: the
    // stage index of a slot visit read back from a76 was the high word of a double; memory access fault at the first launch).  Not a source bug that could be found: the
    // instantiation is retired, those horizons take the workspace variant like the longer ones -- and its straggler hand-over with it)
    h->lat_mem = N > 16 || h->lat_mem_forced;
    if (h->lat_mem) {
        if (!h->d_lat_ws && hipMalloc((void**)&h->d_lat_ws, lat_ws_bytes(cap)) != hipSuccess) { *why = "hipMalloc failed for k_solve_lat's workspace"; return PG_ERR_HIP; }
        C.lat_ws = h->d_lat_ws;
        if (!h->d_lat_spc && hipMalloc((void**)&h->d_lat_spc, lat_spc_bytes((int)cap, N)) != hipSuccess) { *why = "hipMalloc failed for k_solve_lat's stage constants"; return PG_ERR_HIP; }
        C.lat_spc = h->d_lat_spc;
        if (!h->d_hand_r && hipMalloc((void**)&h->d_hand_r, cap * LAT_HAND_R * sizeof(real)) != hipSuccess) { *why = "hipMalloc failed for k_solve_lat's hand-over records"; return PG_ERR_HIP; }
        if (!h->d_hand_i && hipMalloc((void**)&h->d_hand_i, cap * LAT_HAND_I * sizeof(int)) != hipSuccess) { *why = "hipMalloc failed for k_solve_lat's hand-over records"; return PG_ERR_HIP; }
    } else C.lat_ws = nullptr;
    return PG_OK;
}

extern "C" {

int pg_default_config(pg_config* c) {
    if (!c) return PG_ERR_INVALID;
    memset(c, 0, sizeof(*c));
    pg_vehicle& P = c->vehicle;                      // vehicles.jl:1-59
    P.G = 9.80665;
    double mfl = 484, mfr = 455, mrl = 521, mrr = 504;
    P.m = mfl + mfr + mrl + mrr; P.Izz = 2900; P.L = 2.87;
    P.a = (mrl + mrr) / P.m * P.L; P.b = (mfl + mfr) / P.m * P.L;
    P.h = 0.1 * P.b / P.L + 0.1 * P.a / P.L + 0.37;
    P.mu = 0.92; P.Caf = 150e3; P.Car = 220e3; P.Fx_max = 5600; P.Px_max = 75e3; P.Cd0 = 241.0; P.Cd1 = 25.1; P.Cd2 = 0.0;
    P.fwd_frac = 0.0; P.rwd_frac = 1.0 - P.fwd_frac; P.fwb_frac = 0.6; P.rwb_frac = 1.0 - P.fwb_frac;
    double f1 = -P.m * P.G * P.a * P.mu / (P.L * P.rwb_frac + P.mu * P.h), f2 = -P.m * P.G * P.b * P.mu / (P.L * P.fwb_frac - P.mu * P.h);
    P.Fx_min = f1 > f2 ? f1 : f2;
    P.delta_max = 18 * M_PI / 180; P.kappa_max = tan(P.delta_max) / P.L;
    pg_control_params& U = c->control;               // coupled_lat_long.jl:23-38
    U.V_min = 1.0; U.V_max = 15.0; U.k_V = 10.0 / 4 / 100; U.k_s = 10.0 / 4 / 10000; U.deltadot_max = 0.344;
    U.Q_ds = 1.0; U.Q_dpsi = 1.0; U.Q_e = 1.0; U.W_beta = 50 / (10 * M_PI / 180); U.W_r = 50.0; U.W_HJI = 500.0; U.N_HJI = 3;
    U.R_delta = 0.0; U.R_ddelta = 0.1; U.R_Fx = 0.0; U.R_dFx = 0.5;
    c->N_short = 10; c->N_long = 20; c->dt_short = 0.01; c->dt_long = 0.2; c->use_correction_step = 1;   // coupled_lat_long.jl:42-43
    c->rk4_substeps = 10; c->hji_eps = 0.05; c->batch_capacity = 4096; c->device = 0;
    c->ipm_max_iter = 40; c->ipm_mu0 = 100.0; c->formulation = PG_COUPLED; c->walls = 0; c->wall_weight = 1000.0;
    c->polish = 1; c->warm_polish = 1; c->cold_guess = 8;
#ifdef PG_F32
    c->ipm_tol = 1e-5; c->polish_rho = 1e3; c->polish_tol = 1e-4; c->polish_ipm_tol = 1e-4;
#else
    c->ipm_tol = 1e-12; c->polish_rho = 1e7; c->polish_tol = 1e-9; c->polish_ipm_tol = 3e-6;
#endif
    return PG_OK;
}

int pg_default_config_decoupled(pg_config* c) {
    int rc = pg_default_config(c); if (rc) return rc;
    pg_control_params& U = c->control;               // decoupled_lat_long.jl:18-30
    const double d10 = 10 * M_PI / 180;
    U.Q_ds = 0.0; U.Q_dpsi = 1.0 / (d10 * d10); U.Q_e = 1.0; U.W_beta = 50 / d10; U.W_r = 50.0; U.W_HJI = 0.0; U.N_HJI = 0;
    U.R_delta = 0.0; U.R_ddelta = 0.01 / (d10 * d10); U.R_Fx = 0.0; U.R_dFx = 1.0;      // R_dFx only pins the inert Fx slot of the embedding
    c->formulation = PG_DECOUPLED;
    // polish ON (round 3; it used to be off here): with the interior point alone 5 (48 with the wall rows) of the 4096 N = 50 benchmark instances end 1e-6 .. 1e-4 from the
    // exact optimum of their own QP data -- rows that are nearly degenerate, on horizons whose optimum leaves the linearisation by kilometres --; behind the polish of
    // k_solve_lat every one of them is within 1.3e-7 (tests/test_gpu_decoupled.py::test_config5_as_shipped_every_instance_against_the_oracle)
    c->polish = 1;
#ifdef PG_F32
    c->ipm_tol = 1e-4;                               // the ill-conditioned 8 s lateral horizon stalls near 1e-4 in fp32 (tests/test_gpu_f32.py)
#endif
    return PG_OK;
}

static void free_all(pg_handle* h) {
    void* ptrs[] = {h->d_traj, h->d_traj_len, h->d_traj_idx, h->d_in, h->d_out, h->d_solved, h->d_ts, h->d_dt, h->d_prev_ts, h->d_sep, h->d_nodes,
                    h->d_qp, h->d_x7, h->d_vg8, h->d_Mb, h->d_solx, h->d_sigma, h->d_mu, h->d_active, h->d_knots, h->d_hnodes, h->d_hcells, h->d_pol_u2, h->d_pol_u, h->d_pol_src, h->d_ws4, h->d_walls, h->d_mask, h->d_polish, h->d_lam, h->d_todo, h->d_order, h->d_naux, h->d_progress, h->d_lat, h->d_lat_aux, h->d_lat_ws, h->d_tstart, h->d_hand_r, h->d_hand_i, h->d_lat_spc};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    if (h->h_stage) (void)hipHostFree(h->h_stage);
    if (h->sg.x) (void)hipGraphExecDestroy(h->sg.x);
    if (h->sg.g) (void)hipGraphDestroy(h->sg.g);
    if (h->sg.own) (void)hipStreamDestroy(h->sg.own);
    if (h->ev_ok) for (int i = 0; i < 4; i++) (void)hipEventDestroy(h->ev[i]);
}

const char* pg_last_error(const pg_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int pg_create(const pg_config* cfg, pg_handle** out) {
    if (!cfg || !out) return PG_ERR_INVALID;
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) { g_create_error = "no HIP device available (this library has no CPU path)"; return PG_ERR_NO_DEVICE; }
    if (cfg->device < 0 || cfg->device >= ndev) { g_create_error = "device ordinal out of range"; return PG_ERR_INVALID; }
    if (cfg->formulation != PG_COUPLED && cfg->formulation != PG_DECOUPLED) { g_create_error = "unknown formulation"; return PG_ERR_INVALID; }
    if (cfg->walls != 0 && (cfg->walls != 1 || cfg->formulation != PG_DECOUPLED)) { g_create_error = "walls = 1 is an option of the decoupled formulation only"; return PG_ERR_INVALID; }
    if (cfg->polish && !(cfg->polish_rho > 0.0 && cfg->polish_tol > 0.0)) { g_create_error = "polish needs polish_rho > 0 and polish_tol > 0"; return PG_ERR_INVALID; }
    if (cfg->N_short < 1 || cfg->N_long < 0 || cfg->N_short + cfg->N_long + 1 > 64 || cfg->batch_capacity < 1 || cfg->rk4_substeps < 1) {
        g_create_error = "invalid horizon / capacity (need 1 <= N_short, N_short+N_long+1 <= 64)"; return PG_ERR_INVALID;
    }
#ifdef PG_F32
    if (cfg->formulation == PG_DECOUPLED && cfg->N_short + cfg->N_long > 32 && cfg->allow_f32_long_lateral != 1) {
        g_create_error = "the fp32 library refuses the decoupled formulation beyond 32 intervals (applied steering up to 6e-3 rad off on the N = 50 benchmark batch: use libpigeon_hip.so, "
                         "or set pg_config.allow_f32_long_lateral = 1 to run it anyway)";
        return PG_ERR_INVALID;
    }
#endif
    if (hipSetDevice(cfg->device) != hipSuccess) { g_create_error = "hipSetDevice failed"; return PG_ERR_HIP; }
    pg_handle* h = new pg_handle();
    h->cfg = *cfg;
    DevCfg& C = h->dc; memset(&C, 0, sizeof(C));
    fill_dev_params(C, cfg); C.Ns = cfg->N_short; C.Nl = cfg->N_long; C.N = C.Ns + C.Nl; C.NN = C.N + 1;
    C.dt_short = cfg->dt_short; C.dt_long = cfg->dt_long; C.use_correction_step = cfg->use_correction_step; C.nsub = cfg->rk4_substeps;
    C.alias_prev_ts = 1; C.has_hji = 0; C.hji_eps = (real)cfg->hji_eps;
    C.tg_short = jl_scalar_times_unitrange(cfg->dt_short, 0, cfg->N_short); C.tg_long = jl_scalar_times_unitrange(cfg->dt_long, 1, cfg->N_long); C.time_grid_naive = 0;
    C.un0 = (real)cfg->vehicle.delta_max; C.un1 = (real)fmax(-cfg->vehicle.Fx_min, cfg->vehicle.Fx_max);      // coupled_lat_long.jl:199
    C.fxmin_n = (real)(cfg->vehicle.Fx_min / fmax(-cfg->vehicle.Fx_min, cfg->vehicle.Fx_max));
    C.formulation = cfg->formulation; C.ux_dummy = (real)(0.5 * (cfg->control.V_min + cfg->control.V_max));
    C.dbg_poison = 0; C.dbg_instance = -1;                              // (-DPG_DIAG builds: option "diag_instance")
    if (cfg->formulation == PG_DECOUPLED) {          // no u normalisation in the lateral QP (decoupled_lat_long.jl:134-226); inert slots pinned
        C.un0 = 1.0; C.un1 = 1.0; C.fxmin_n = -1.0;
        C.cp.Q_ds = 0.0; C.cp.R_Fx = 0.0; C.cp.R_dFx = 1.0; C.cp.N_HJI = 0; C.cp.W_HJI = 0.0;
    }
    C.qp_len = 84 * C.N + 11;
    C.ipm_max_iter = cfg->ipm_max_iter; C.ipm_tol = (real)cfg->ipm_tol; C.ipm_mu0 = (real)cfg->ipm_mu0;
    C.polish = cfg->polish != 0; C.polish_rho = (real)cfg->polish_rho; C.polish_tol = (real)cfg->polish_tol; C.polish_ipm_tol = (real)cfg->polish_ipm_tol; C.warm_polish = cfg->warm_polish != 0; C.cold_guess = cfg->cold_guess > 0 ? cfg->cold_guess : 0;
    const size_t cap = (size_t)cfg->batch_capacity; const int N = C.N, NN = C.NN;
#define ALLOC(ptr, count, type) do { if (hipMalloc((void**)&(ptr), (size_t)(count) * sizeof(type)) != hipSuccess) { g_create_error = "hipMalloc failed for " #ptr; free_all(h); delete h; return PG_ERR_HIP; } } while (0)
    h->in_dbl_off = (cap * 13 * sizeof(real) + 7) / 8 * 8; h->in_bytes = h->in_dbl_off + cap * 16; h->out_bytes = cap * (3 * sizeof(real) + 8);
    ALLOC(h->d_in, h->in_bytes, char); ALLOC(h->d_out, h->out_bytes, char);
    h->d_state = (real*)h->d_in; h->d_control = h->d_state + cap * 6; h->d_other = h->d_control + cap * 3; h->d_t0 = (double*)(h->d_in + h->in_dbl_off); h->d_toff = h->d_t0 + cap;
    h->d_u = (real*)h->d_out; h->d_status = (int*)(h->d_u + cap * 3); h->d_iters = h->d_status + cap;
    ALLOC(h->d_tstart, cap, double);
    ALLOC(h->d_solved, 2 * cap, int); h->d_wfail = h->d_solved + cap; ALLOC(h->d_mask, cap, uint8_t); ALLOC(h->d_ts, cap * NN, double); ALLOC(h->d_dt, cap * N, double); ALLOC(h->d_prev_ts, cap * NN, double);
    ALLOC(h->d_sep, cap * 4, real); ALLOC(h->d_nodes, cap * NN * 10, real); ALLOC(h->d_qp, cap * C.qp_len, real);
    ALLOC(h->d_x7, cap * 7, real); ALLOC(h->d_vg8, cap * 8, real); ALLOC(h->d_Mb, cap * 4, real);
    ALLOC(h->d_solx, cap * NN * 8, real); ALLOC(h->d_sigma, cap * N * 3, real); ALLOC(h->d_mu, cap, real);
    ALLOC(h->d_polish, cap, int); ALLOC(h->d_todo, cap + 8, int); ALLOC(h->d_lam, cap * N * 16, real); ALLOC(h->d_order, 2 * cap + 2, int); ALLOC(h->d_naux, cap * NN * 4, real); ALLOC(h->d_progress, cap / NODES_IPB + 8, int); ALLOC(h->d_active, cap * N, uint16_t);
#ifdef PG_EXPERIMENTAL_SOLVE4
    { const char* e = getenv("PG_SOLVER"); h->solve_quad = N <= 32 && e && strcmp(e, "quad") == 0; }   // experimental four-instances-per-wavefront kernel (experimental/pg_solve4.hip)
    if (h->solve_quad) ALLOC(h->d_ws4, cap * ws4_len(N), real);
#endif
    if (cfg->walls) { ALLOC(h->d_walls, cap * N * 2, real); C.walls = 1; C.wall_weight = (real)cfg->wall_weight; C.wall_edges = h->d_walls; if (h->solve_quad) { g_create_error = "the experimental quad solver does not carry the wall rows"; free_all(h); delete h; return PG_ERR_INVALID; } }
    ALLOC(h->d_pol_u2, cap * 2, real); ALLOC(h->d_pol_u, cap * 3, real); ALLOC(h->d_pol_src, cap, int);
    // The lateral formulation has a solve kernel of its own (k_solve_lat, pg_solve_lat.hip) for horizons beyond 20 intervals.  Shorter ones -- the reference's on-vehicle
    // N_short = 5, N_long = 10 among them -- stay with the embedding in k_solve when the polish is on: its active-set rounds from the empty set serve a short lateral QP
    // without any interior-point iteration (N = 15, 4096 instances: 0.12 ms against 0.37), an advantage that is gone at N = 30 (2.4 against 1.5 ms) and reversed at
    // N = 50 (12.6 against 5.9).  Option "lateral_solver" = 1 / 2 forces the choice (configure_lateral below).
    if (cfg->formulation == PG_DECOUPLED) {
        // defaults of k_solve_lat's tuning options (pg_set_option)
        C.lat_mu0_cost = real(10.0); C.lat_pin = 1; C.lat_polish2 = 1; C.lat_far_cost = real(3e4); C.lat_rho_scale = sizeof(real) == 8 ? real(1e3) : real(1.0); C.lat_polish_rounds = 3; C.lat_settle = 0;
        C.lat_warm_rounds = 2;      /* (2 since the two-launch warm step: an attempt that needs a third working set is cheaper to hand to the cold list: 2.03 -> 1.95 ms, 2.9 -> 2.7 with walls) */
        C.lat_wipm = 0; C.lat_wmu = real(1e-2); C.lat_wtau = real(1e-4); C.lat_aux_gate = 1;
        std::string why;
        if (configure_lateral(h, &why) != PG_OK) { g_create_error = why; free_all(h); delete h; return PG_ERR_HIP; }
    }
#undef ALLOC
    h->stage_bytes = h->in_bytes > h->out_bytes ? h->in_bytes : h->out_bytes;          // inputs: state 6 + control 3 + other 4 (real) + t0 + time_offset (double); outputs reuse the front of it
    if (hipHostMalloc((void**)&h->h_stage, h->stage_bytes, hipHostMallocDefault) != hipSuccess) { g_create_error = "hipHostMalloc failed for the staging buffer"; free_all(h); delete h; return PG_ERR_HIP; }
    // initial ts = 1..NN (model_predictive_control.jl:13), solved = false
    {
        std::vector<double> ts(cap * NN), dt(cap * N, 1.0);
        for (size_t b = 0; b < cap; b++) for (int i = 0; i < NN; i++) ts[b * NN + i] = i + 1;
        (void)hipMemcpy(h->d_ts, ts.data(), ts.size() * 8, hipMemcpyHostToDevice);
        (void)hipMemcpy(h->d_prev_ts, ts.data(), ts.size() * 8, hipMemcpyHostToDevice);
        (void)hipMemcpy(h->d_dt, dt.data(), dt.size() * 8, hipMemcpyHostToDevice);
        (void)hipMemset(h->d_solved, 0, 2 * cap * sizeof(int));
        (void)hipMemset(h->d_progress, 0, (cap / NODES_IPB + 8) * sizeof(int));
        (void)hipMemset(h->d_status, 0, cap * sizeof(int));
        (void)hipMemset(h->d_other, 0, cap * 4 * sizeof(real));
        (void)hipMemset(h->d_solx, 0, cap * NN * 8 * sizeof(real));
        (void)hipMemset(h->d_lam, 0, cap * N * 16 * sizeof(real));          // pg_get_multipliers promises 0 off the working set: the kernels write only the rows they own
        (void)hipMemset(h->d_active, 0, cap * N * sizeof(uint16_t));
        (void)hipMemset(h->d_todo + cap, 0, 8 * sizeof(int));
        if (hipDeviceSynchronize() != hipSuccess) { g_create_error = "initial fills failed"; free_all(h); delete h; return PG_ERR_HIP; }   // hipMemset may return before the fill has run
    }
    for (int i = 0; i < 4; i++) if (hipEventCreate(&h->ev[i]) != hipSuccess) { g_create_error = "hipEventCreate failed"; free_all(h); delete h; return PG_ERR_HIP; }
    h->ev_ok = true;
    // defaults of the build-defined options (pg_set_option changes them per handle; nothing here reads the process environment)
    C.hji_seed = 0; C.clip_guess = 1; C.clip_stops = 0; C.ck_riccati = 1; C.warm_trivial_cold = 1; C.hji_rounds = 0;
    // horizons up to 32 intervals keep their dynamics blocks resident in LDS (one pass over the QP data); longer ones stream them through a 4-slot ring
    h->solve_ring = N > 32;
#ifdef PG_EXPERIMENTAL_SOLVE4
    h->solve4_lds = lds4_bytes(N);
#endif
    h->solve_lds = (size_t)((h->solve_ring ? 4 : N) * SB + 10 * NN + 8 * NN + 2 * N + 2 * N + 16 * N + 4 * N + 8 * N + 2 * N + 8 * NN + 2 * N + 72 + 100 + 8 + 72 + 4 + 32 + 2 * N + 11 * N) * sizeof(real);
    if (h->solve_lds > 160 * 1024) { g_create_error = "horizon too long for LDS staging"; free_all(h); delete h; return PG_ERR_INVALID; }
    // hipFuncSetAttribute applies to the CURRENT DEVICE's copy of a kernel, and a later handle with a shorter horizon must not lower the limit an earlier handle of the same
    // device relies on: the largest size asked for so far is kept per device ordinal
    static size_t lds_attr_max[64] = {0};
    const int dev_slot = cfg->device < 64 ? cfg->device : 63;
    if (lds_attr_max[dev_slot] < 48 * 1024) lds_attr_max[dev_slot] = 48 * 1024;
    if (h->solve_lds > lds_attr_max[dev_slot])
    {
        lds_attr_max[dev_slot] = h->solve_lds;
        (void)hipFuncSetAttribute((const void*)k_solve<false, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->solve_lds);
        (void)hipFuncSetAttribute((const void*)k_solve<false, false, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->solve_lds);
        (void)hipFuncSetAttribute((const void*)k_solve<false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->solve_lds);
        (void)hipFuncSetAttribute((const void*)k_solve<false, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->solve_lds);
#ifdef PG_DIAG      // (the PROF instantiations exist in the diagnostic library only: pg_debug_solve_cycles)
        (void)hipFuncSetAttribute((const void*)k_solve<true, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->solve_lds);
        (void)hipFuncSetAttribute((const void*)k_solve<true, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->solve_lds);
#endif
    }
    h->lat_lds = lat_lds_doubles(N) * sizeof(real);
    // (N + 1 <= 64 nodes: at most 4 x 63 x 24 doubles + 72 = 48,960 B, below the 64 KB a launch may ask for without raising the function attribute)
    if (h->solve_lat && h->lat_lds > 64 * 1024) { g_create_error = "horizon too long for k_solve_lat's LDS staging"; free_all(h); delete h; return PG_ERR_INVALID; }
    *out = h;
    return PG_OK;
}

int pg_destroy(pg_handle* h) { if (!h) return PG_ERR_INVALID; (void)hipSetDevice(h->cfg.device); free_all(h); delete h; return PG_OK; }
int pg_get_config(const pg_handle* h, pg_config* out) { if (!h || !out) return PG_ERR_INVALID; *out = h->cfg; return PG_OK; }
int pg_get_u_normalization(const pg_handle* h, double out[2]) {
    if (!h || !out) return PG_ERR_INVALID;
    const bool dec = h->cfg.formulation == PG_DECOUPLED;      // reported in full precision (the host de-normalises with it)
    out[0] = dec ? 1.0 : h->cfg.vehicle.delta_max; out[1] = dec ? 1.0 : fmax(-h->cfg.vehicle.Fx_min, h->cfg.vehicle.Fx_max);
    return PG_OK;
}
int pg_precision_bits(void) { return (int)(8 * sizeof(real)); }
// Layout of the configuration structs as THIS library was compiled: the ctypes mirror (pigeon.jl_amd/_lib.py) and the Julia mirror (julia/PigeonMI355X.jl)
// are hand-written copies of include/pigeon_mpc.h; they compare their own sizeof / field offsets with these numbers before the first pg_create.
int pg_abi_layout(int32_t* out, int32_t n) {
#define OFF(f) (int32_t)offsetof(pg_config, f)
    const int32_t v[] = {(int32_t)sizeof(pg_config), (int32_t)sizeof(pg_vehicle), (int32_t)sizeof(pg_control_params), OFF(control), OFF(N_short), OFF(dt_short), OFF(use_correction_step),
                         OFF(hji_eps), OFF(batch_capacity), OFF(ipm_max_iter), OFF(formulation), OFF(ipm_tol), OFF(ipm_mu0), OFF(walls), OFF(wall_weight), OFF(polish), OFF(polish_rho),
                         OFF(polish_tol), OFF(polish_ipm_tol), OFF(warm_polish), OFF(cold_guess), (int32_t)offsetof(pg_control_params, N_HJI), (int32_t)offsetof(pg_vehicle, kappa_max)};
#undef OFF
    const int32_t cnt = (int32_t)(sizeof(v) / sizeof(v[0]));
    if (out) for (int32_t i = 0; i < cnt && i < n; i++) out[i] = v[i];
    return cnt;
}
int pg_qp_len(const pg_handle* h) { return h ? h->dc.qp_len : PG_ERR_INVALID; }
int pg_set_stream(pg_handle* h, void* s) {
    if (!h) return PG_ERR_INVALID;
    // pg_set_inputs / pg_step leave asynchronous copies from (and into) the ONE pinned staging buffer queued on the old stream: they must have landed before kernels on
    // the new stream read the inputs or the next call overwrites the buffer
    if ((hipStream_t)s != h->stream) { (void)hipSetDevice(h->cfg.device); HIPCHK(h, hipStreamSynchronize(h->stream)); }
    h->stream = (hipStream_t)s;
    return PG_OK;
}
int pg_set_pipeline(pg_handle* h, int32_t mode) { if (!h || mode < 0 || mode > 1) return PG_ERR_INVALID; h->pipeline = mode; return PG_OK; }
int pg_set_fusion(pg_handle* h, int32_t mode) { if (!h || mode < 0 || mode > 2) return PG_ERR_INVALID; h->fuse = mode; return PG_OK; }
// ---- build-defined options by name (include/pigeon_mpc.h lists them).  One table serves pg_set_option and pg_get_option. ----
namespace {
struct OptRef { int* i; real* r; int64_t* stat; double lo, hi; bool integer; };
static bool find_option(pg_handle* h, const char* name, OptRef* o) {
    DevCfg& C = h->dc;
    *o = OptRef{nullptr, nullptr, nullptr, 0.0, 0.0, true};
    auto I = [&](int* p, double lo, double hi) { o->i = p; o->lo = lo; o->hi = hi; o->integer = true; return true; };
    auto R = [&](real* p, double lo, double hi) { o->r = p; o->lo = lo; o->hi = hi; o->integer = false; return true; };
    auto S = [&](int64_t* p) { o->stat = p; return true; };
    const std::string n(name);
    // k_solve (coupled QP)
    if (n == "clip_guess") return I(&C.clip_guess, 0, 1);
    if (n == "clip_stops") return I(&C.clip_stops, 0, 1);
    if (n == "ck_riccati") return I(&C.ck_riccati, 0, 1);
    if (n == "warm_trivial_cold") return I(&C.warm_trivial_cold, 0, 1);
    if (n == "hji_seed") return I(&C.hji_seed, 0, 4);
    if (n == "hji_rounds") return I(&C.hji_rounds, 0, 64);
    if (n == "solve_split") return I(&h->split_solve, 0, 1);
    // launch shape
    if (n == "pipe_min") return I(&h->pipe_min, 0, 1 << 30);
    if (n == "pipe_first") return I(&h->pipe_first, 0, 64);
    if (n == "pipe_pub_short") return I(&h->pipe_pub_short, 1, 64);
    if (n == "pipe_pub_long") return I(&h->pipe_pub_long, 1, 64);
    if (n == "pipe_max") return I(&h->pipe_max, 0, 256 * NODES_IPB);
    if (n == "lin_lanes") return I(&h->lin_lpi, 1, 2);
    if (n == "graph") return I(&h->graph_mode, 0, 1);
    if (n == "phase_timing") return I(&h->phase_timing, 0, 1);
    if (n == "hji_cell_dims") return I(&h->hji_cell_dims, 3, 7);
    if (n == "time_grid_naive") return I(&C.time_grid_naive, 0, 1);
    // k_solve_lat (lateral QP)
    if (n == "lateral_solver") return I(&h->lateral_solver, 0, 2);
    if (n == "lat_split") return I(&h->split_lat, 0, 1);
    if (n == "lat_handover") return I(&h->lat_handover, 0, 1);
    if (n == "nodes_serial") return I(&C.nodes_serial, 0, 1);
    if (n == "lat_aux_gate") return I(&C.lat_aux_gate, 0, 1);
    if (n == "lat_hand_target") return I(&h->lat_hand_target, 0, 1 << 30);
    if (n == "lat_hand_min") return I(&h->lat_hand_min, 1, 1 << 20);
    if (n == "lat_hand_cap") return I(&h->lat_hand_cap, 0, 1 << 20);
    if (n == "lat_hand_work") return I(&h->lat_hand_work, 0, 1 << 20);
    if (n == "lat_hand_w0") return I(&h->lat_hand_w0, 0, 1 << 10);
    if (n == "lat_hand_batch") return I(&h->lat_hand_batch, 1, 1 << 30);
    if (n == "lat_single_max") return I(&h->lat_single_max, 0, 1 << 30);
    if (n == "lat_pack_only") return I(&h->lat_pack_only, 0, 1);
    if (n == "lat_polish2") return I(&C.lat_polish2, 0, 1);
    if (n == "lat_pin") return I(&C.lat_pin, 0, 1);
    if (n == "lat_polish_rounds") return I(&C.lat_polish_rounds, 2, 64);
    if (n == "lat_settle") return I(&C.lat_settle, 0, 2);
    if (n == "lat_warm_rounds") return I(&C.lat_warm_rounds, 0, 64);
    if (n == "lat_wipm") return I(&C.lat_wipm, 0, 1);
    if (n == "lat_mu0_cost") return R(&C.lat_mu0_cost, 1e-300, 1e300);
    if (n == "lat_far_cost") return R(&C.lat_far_cost, 1e-300, 1e300);
    if (n == "lat_rho_scale") return R(&C.lat_rho_scale, 1e-300, 1e12);
    if (n == "lat_wmu") return R(&C.lat_wmu, 1e-300, 1e300);
    if (n == "lat_wtau") return R(&C.lat_wtau, 1e-300, 1e300);
    // read-only launch statistics of this handle
    if (n == "stat_pipelined_launches") return S(&h->stat_pipelined);
    if (n == "stat_split_solve_launches") return S(&h->stat_split);
    if (n == "stat_single_solve_launches") return S(&h->stat_single);
    if (n == "stat_lat_two_launch_solves") return S(&h->stat_lat_two);
    if (n == "stat_lat_handover_solves") return S(&h->stat_lat_hand);
    if (n == "stat_lat_one_per_wavefront_solves") return S(&h->stat_lat_single);
#ifdef PG_DIAG      // diagnostic build only (libpigeon_hip_diag.so): fault injection and traces have no place in the shipped libraries
    if (n == "diag_instance") return I(&C.dbg_instance, -1, 1 << 30);
    if (n == "diag_lat_poison") return I(&C.dbg_poison, 0, 1);
    if (n == "diag_lin_groups") return I(&h->lin_groups, 0, 8);
    if (n == "diag_timeline") return I(&h->debug_timeline, 0, 1);
#endif
    return false;
}
}  // namespace
int pg_set_option(pg_handle* h, const char* name, double value) {
    if (!h || !name) return PG_ERR_INVALID;
#ifdef PG_DIAG
    if (strcmp(name, "diag_pipe_fault") == 0) { h->pipe_fault = value != 0.0; return PG_OK; }
#endif
    if (strcmp(name, "lat_workspace") == 0 || strcmp(name, "lateral_solver") == 0) {      // these two decide buffers: applied through configure_lateral, with the stream drained
        REQUIRE(h, h->cfg.formulation == PG_DECOUPLED, "pg_set_option: an option of the decoupled formulation");
        const bool which_solver = strcmp(name, "lateral_solver") == 0;
        REQUIRE(h, value == 0.0 || value == 1.0 || (value == 2.0 && which_solver), "pg_set_option: value out of range");
        HIPCHK(h, hipSetDevice(h->cfg.device)); HIPCHK(h, hipStreamSynchronize(h->stream));
        // (validated BEFORE anything changes, and a configure_lateral that fails -- an allocation -- leaves the handle as it was: ADVICE r5)
        if (which_solver && value == 1.0 && lat_lds_doubles(h->dc.N) * sizeof(real) > 64 * 1024) { h->err = "horizon too long for k_solve_lat's LDS staging"; return PG_ERR_INVALID; }
        const int old_solver = h->lateral_solver; const bool old_forced = h->lat_mem_forced, old_lat = h->solve_lat;
        if (which_solver) h->lateral_solver = (int)value; else h->lat_mem_forced = value != 0.0;
        std::string why;
        if (configure_lateral(h, &why) != PG_OK) {
            h->lateral_solver = old_solver; h->lat_mem_forced = old_forced; std::string why2; (void)configure_lateral(h, &why2);
            h->err = why; return PG_ERR_HIP;
        }
        // the two solve kernels read different QP records (k_solve: the embedded block, k_solve_lat: the packed stage records) and the last update_QP! wrote what the OLD
        // choice needed: pg_solve refuses until pg_update_qp (or a whole-batch pg_set_qp) has run again
        if (h->solve_lat != old_lat) h->qp_stale = true;
        return PG_OK;
    }
    OptRef o;
    if (!find_option(h, name, &o)) { h->err = std::string("pg_set_option: unknown option '") + name + "'"; return PG_ERR_INVALID; }
    if (o.stat) { h->err = std::string("pg_set_option: '") + name + "' is read-only"; return PG_ERR_INVALID; }
    if (!(value >= o.lo && value <= o.hi) || (o.integer && value != (double)(long long)value)) { h->err = std::string("pg_set_option: value out of range for '") + name + "'"; return PG_ERR_INVALID; }
    if (strcmp(name, "hji_cell_dims") == 0 && !(value == 3 || value == 5 || value == 7)) { h->err = "pg_set_option: hji_cell_dims is 3, 5 or 7"; return PG_ERR_INVALID; }
    if (o.i) *o.i = (int)value; else *o.r = (real)value;
    return PG_OK;
}
int pg_get_option(pg_handle* h, const char* name, double* value) {
    if (!h || !name || !value) return PG_ERR_INVALID;
#ifdef PG_DIAG
    if (strcmp(name, "diag_pipe_fault") == 0) { *value = h->pipe_fault ? 1.0 : 0.0; return PG_OK; }
#endif
    if (strcmp(name, "lat_workspace") == 0) { *value = h->lat_mem ? 1.0 : 0.0; return PG_OK; }
    if (strcmp(name, "stat_whole_batch_solves") == 0) {      // counted ON THE DEVICE by the full k_solve whenever it takes the whole batch (SolveOut::mode): drains the stream
        HIPCHK(h, hipSetDevice(h->cfg.device)); HIPCHK(h, hipStreamSynchronize(h->stream));
        int v = 0; HIPCHK(h, hipMemcpy(&v, h->d_todo + (size_t)h->cfg.batch_capacity + 5, sizeof(int), hipMemcpyDeviceToHost));
        *value = (double)v; return PG_OK;
    }
    if (strcmp(name, "lateral_solver_in_use") == 0) { *value = h->solve_lat ? 1.0 : (h->cfg.formulation == PG_DECOUPLED ? 2.0 : 0.0); return PG_OK; }
    OptRef o;
    if (!find_option(h, name, &o)) { h->err = std::string("pg_get_option: unknown option '") + name + "'"; return PG_ERR_INVALID; }
    *value = o.stat ? (double)*o.stat : (o.i ? (double)*o.i : (double)*o.r);
    return PG_OK;
}
int pg_get_pipeline_fallbacks(pg_handle* h, int64_t* count) {
    if (!h || !count) return PG_ERR_INVALID;
    HIPCHK(h, hipSetDevice(h->cfg.device));                 // (a multi-GPU process: the copy below must not depend on whichever device happens to be current)
    HIPCHK(h, hipStreamSynchronize(h->stream));
    int v = 0;
    HIPCHK(h, hipMemcpy(&v, h->d_progress + (size_t)h->cfg.batch_capacity / NODES_IPB + 7, sizeof(int), hipMemcpyDeviceToHost));
    // the device word is a 32-bit counter of waiting wavefronts that gave up: the host keeps the 64-bit total and folds the device word into it (a wrap of the 32-bit
    // word between two calls would need 2^31 fall-backs -- at one per 20 ms wait, more than a year of nothing but fall-backs)
    h->fallback_total += (int64_t)(uint32_t)((uint32_t)v - (uint32_t)h->fallback_seen);
    h->fallback_seen = v;
    *count = h->fallback_total;
    return PG_OK;
}
int pg_synchronize(pg_handle* h) { if (!h) return PG_ERR_INVALID; HIPCHK(h, hipStreamSynchronize(h->stream)); return PG_OK; }

// install a library: channels[n_traj][10][Lmax] (t, s, V, A, E, N, psi, kappa, edge_L, edge_R), L[k] valid nodes of trajectory k
static int install_trajectories(pg_handle* h, int n_traj, int Lmax, const int32_t* L, const double* channels) {
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->d_traj) { (void)hipFree(h->d_traj); h->d_traj = nullptr; }
    if (h->d_traj_len) { (void)hipFree(h->d_traj_len); h->d_traj_len = nullptr; }
    if (h->d_traj_idx) { (void)hipFree(h->d_traj_idx); h->d_traj_idx = nullptr; }
    h->traj_idx_B = 0;
    const size_t stride = (size_t)10 * Lmax;
    HIPCHK(h, hipMalloc((void**)&h->d_traj, (size_t)n_traj * stride * sizeof(real)));
    { int rc = up(h, h->d_traj, channels, (size_t)n_traj * stride); if (rc) return rc; }
    HIPCHK(h, hipMalloc((void**)&h->d_traj_len, (size_t)n_traj * sizeof(int)));
    HIPCHK(h, hipMemcpy(h->d_traj_len, L, (size_t)n_traj * sizeof(int), hipMemcpyHostToDevice));
    h->traj_t_end = channels[(size_t)L[0] - 1];          // trajectory.t[end] of trajectory 0 (channel 0 = t): the stop of pg_simulate_dev's clock range
    TrajView& T = h->dc.traj; T.L = L[0];
    const real* p = h->d_traj; const size_t c = (size_t)Lmax;
    T.t = p; T.s = p + c; T.V = p + 2 * c; T.A = p + 3 * c; T.E = p + 4 * c; T.N = p + 5 * c; T.psi = p + 6 * c; T.kappa = p + 7 * c; T.edge_L = p + 8 * c; T.edge_R = p + 9 * c;
    h->dc.n_traj = n_traj; h->dc.traj_stride = (long)stride; h->dc.traj_len = h->d_traj_len; h->dc.traj_idx = nullptr;
    h->traj_L = L[0];
    return PG_OK;
}

int pg_set_trajectory(pg_handle* h, int32_t L, const double* t, const double* s, const double* V, const double* A, const double* E, const double* N,
                      const double* psi, const double* kappa, const double* theta, const double* phi, const double* edge_L, const double* edge_R) {
    if (!h) return PG_ERR_INVALID;
    REQUIRE(h, L >= 2 && t && s && V && A && E && N && psi && kappa, "pg_set_trajectory: need L >= 2 and the eight read channels");
    (void)theta; (void)phi;                                   // carried by TrajectoryTube (trajectories.jl:17-18) but read by nothing on this path
    std::vector<double> pack((size_t)10 * L);
    const double* src[8] = {t, s, V, A, E, N, psi, kappa};
    for (int k = 0; k < 8; k++) memcpy(pack.data() + (size_t)k * L, src[k], (size_t)L * 8);
    for (int i = 0; i < L; i++) { pack[(size_t)8 * L + i] = edge_L ? edge_L[i] : 4.0; pack[(size_t)9 * L + i] = edge_R ? edge_R[i] : -4.0; }   // defaults: trajectories.jl:42
    const int32_t Ls = L;
    return install_trajectories(h, 1, L, &Ls, pack.data());
}

int pg_set_trajectories(pg_handle* h, int32_t n_traj, int32_t Lmax, const int32_t* L, const double* channels) {
    if (!h) return PG_ERR_INVALID;
    REQUIRE(h, n_traj >= 1 && Lmax >= 2 && L && channels, "pg_set_trajectories: need n_traj >= 1, Lmax >= 2, lengths and channels");
    for (int k = 0; k < n_traj; k++) REQUIRE(h, L[k] >= 2 && L[k] <= Lmax, "pg_set_trajectories: every trajectory needs 2 <= L[k] <= Lmax");
    return install_trajectories(h, n_traj, Lmax, L, channels);
}

int pg_set_trajectory_index(pg_handle* h, int32_t B, const int32_t* index) {
    if (!h) return PG_ERR_INVALID;
    REQUIRE(h, h->d_traj, "pg_set_trajectory_index: no trajectory library installed");
    REQUIRE(h, B >= 1 && B <= h->cfg.batch_capacity && index, "pg_set_trajectory_index: need 1 <= B <= batch_capacity and an index array");
    for (int b = 0; b < B; b++) REQUIRE(h, index[b] >= 0 && index[b] < h->dc.n_traj, "pg_set_trajectory_index: index out of range of the installed library");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (!h->d_traj_idx) HIPCHK(h, hipMalloc((void**)&h->d_traj_idx, (size_t)h->cfg.batch_capacity * sizeof(int)));
    HIPCHK(h, hipMemcpy(h->d_traj_idx, index, (size_t)B * sizeof(int), hipMemcpyHostToDevice));
    h->dc.traj_idx = h->d_traj_idx; h->traj_idx_B = B;
    return PG_OK;
}

int pg_clear_hji_grid(pg_handle* h) {
    if (!h) return PG_ERR_INVALID;
    (void)hipStreamSynchronize(h->stream);
    if (h->d_knots) (void)hipFree(h->d_knots);
    if (h->d_hnodes) (void)hipFree(h->d_hnodes);
    if (h->d_hcells) (void)hipFree(h->d_hcells);
    h->d_knots = nullptr; h->d_hnodes = nullptr; h->d_hcells = nullptr; h->has_hji = false; h->dc.has_hji = 0;
    return PG_OK;
}

int pg_set_hji_grid(pg_handle* h, const int32_t dims[7], const float* knots_concat, const float* V, const float* gradV) {
    if (!h) return PG_ERR_INVALID;
    REQUIRE(h, dims && knots_concat && V && gradV, "pg_set_hji_grid: null argument");
    REQUIRE(h, h->dc.formulation == PG_COUPLED, "the HJI safety row belongs to the coupled formulation only (coupled_lat_long.jl:341-346)");
    size_t n = 1; int nk = 0;
    for (int d = 0; d < 7; d++) { REQUIRE(h, dims[d] >= 2, "pg_set_hji_grid: every dimension needs >= 2 knots"); n *= dims[d]; nk += dims[d]; }
    HIPCHK(h, hipSetDevice(h->cfg.device));
    pg_clear_hji_grid(h);
    // interleave (V, gradV[7]) into 8-float node records: one 32 B aligned read per corner instead of a 4 B + a 28 B unaligned one
    std::vector<float> rec(n * 8);
    for (size_t i = 0; i < n; i++) { rec[8 * i] = V[i]; for (int k = 0; k < 7; k++) rec[8 * i + 1 + k] = gradV[7 * i + k]; }
    HIPCHK(h, hipMalloc((void**)&h->d_hnodes, rec.size() * sizeof(float)));
    HIPCHK(h, hipMalloc((void**)&h->d_knots, (size_t)nk * sizeof(float)));
    HIPCHK(h, hipMemcpy(h->d_hnodes, rec.data(), rec.size() * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->d_knots, knots_concat, (size_t)nk * sizeof(float), hipMemcpyHostToDevice));
    long st = 1; int ko = 0;
    for (int d = 0; d < 7; d++) { h->hv.dims[d] = dims[d]; h->hv.koff[d] = ko; h->hv.stride[d] = st; st *= dims[d]; ko += dims[d]; }
    h->hv.knots = h->d_knots; h->hv.nodes = h->d_hnodes;
    // cell records (8 corners of dims 1..3 contiguous): built on the device from the compact node records, which are then released
    size_t free_b = 0, total_b = 0;
    HIPCHK(h, hipMemGetInfo(&free_b, &total_b));
    // Device layout: 256 B cell records (cdims = 3: sixteen per lookup; 6 x the node table -- 1.9 GB for the 13 x 13 x 9^5 grid).  Since the lookup reads every record 256
    // contiguous bytes per instruction and lane group (round 4), the three layouts stream at the same rate -- 0.77 / 0.76 / 0.69-0.74 of the HBM peak for 256 B / 1 KiB /
    // 4 KiB records -- so the smallest table is the default (rounds 1-3: 4 KiB records, 19.3 GB).  Option "hji_cell_dims" = 5 / 7 selects the larger records (bench, tests).
    int cd = h->hji_cell_dims; long ncell = 0;
    for (;; cd -= 2) {        // (a larger record asked for must fit a quarter of the free HBM)
        long cs = 1; ncell = 1;
        for (int d = 0; d < 7; d++) { int ext = d < cd ? dims[d] - 1 : dims[d]; h->hv.cstride[d] = cs; cs *= ext; ncell *= ext; }
        if (cd == 3 || ((size_t)ncell << cd) * 32 <= free_b / 4) break;
    }
    h->hv.cdims = cd;
    HIPCHK(h, hipMalloc((void**)&h->d_hcells, ((size_t)ncell << cd) * 8 * sizeof(float)));
    h->hv.cells = h->d_hcells;
    hipLaunchKernelGGL(k_hji_build_cells, dim3((unsigned)((((size_t)ncell << cd) + 255) / 256)), dim3(256), 0, h->stream, h->hv, ncell, h->d_hcells);
    LAUNCH_CHECK(h);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    (void)hipFree(h->d_hnodes); h->d_hnodes = nullptr; h->hv.nodes = nullptr;
    h->has_hji = true; h->dc.has_hji = 1;
    return PG_OK;
}

__global__ void k_reset(int B, const uint8_t* mask, int* solved, int* wfail) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B && (!mask || mask[b])) { solved[b] = 0; wfail[b] = 0; }
}

int pg_reset(pg_handle* h, const uint8_t* mask) {
    if (!h) return PG_ERR_INVALID;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const int cap = h->cfg.batch_capacity;
    h->warm_B = 0; h->order_B = 0;
    if (!mask) { HIPCHK(h, hipMemsetAsync(h->d_solved, 0, (size_t)2 * cap * sizeof(int), h->stream)); return PG_OK; }      // (solved flags and the back-off words behind them: one fill)
    REQUIRE(h, h->B > 0, "pg_reset with a mask needs inputs installed (B known)");
    HIPCHK(h, hipMemcpyAsync(h->d_mask, mask, h->B, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_reset, dim3((h->B + 255) / 256), dim3(256), 0, h->stream, h->B, h->d_mask, h->d_solved, h->d_wfail);
    LAUNCH_CHECK(h);
    HIPCHK(h, hipStreamSynchronize(h->stream));            // the caller may reuse `mask`
    return PG_OK;
}

static void stage_block(pg_handle* h, int32_t B, const double* s_, const double* c_, const double* t0, const double* o_, const double* toff);
static int set_inputs(pg_handle* h, int32_t B, const void* state, const void* control, const double* t0, const void* other, const double* toff, bool host) {
    if (!h) return PG_ERR_INVALID;
    REQUIRE(h, B >= 1 && B <= h->cfg.batch_capacity, "batch size outside [1, batch_capacity]");
    REQUIRE(h, state && control && t0, "state, control and t0 are required");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    h->B = B; h->sim_idx = 0;      // (new times: the clock of pg_simulate_dev restarts from them)
    const hipMemcpyKind kind = host ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
    int rc;
    if (host) {
        // the caller's arrays are converted / copied into the pinned staging buffer here and now (so they may be reused as soon as this returns) and travel
        // from there with asynchronous copies ordered before the kernels on the handle's stream: no synchronisation on the way in.  A previous call's
        // copies must have left the buffer first.
        HIPCHK(h, hipStreamSynchronize(h->stream));
        const double* s_ = (const double*)state; const double* c_ = (const double*)control; const double* o_ = (const double*)other;
        if (B == h->cfg.batch_capacity) {
            // the batch fills the handle: the staging buffer takes the layout of the device block and ONE copy carries all five arrays (a single controller at
            // 100 Hz -- B = capacity = 1 -- used to pay five copies and two memsets of a few bytes each per step)
            stage_block(h, B, s_, c_, t0, o_, toff);
            HIPCHK(h, hipMemcpyAsync(h->d_in, h->h_stage, h->in_bytes, kind, h->stream));
            return PG_OK;
        }
        real* st = (real*)h->h_stage; real* ct = st + (size_t)B * 6; real* ot = ct + (size_t)B * 3; double* tt = (double*)(h->h_stage + ((size_t)B * 13 * sizeof(real) + 7) / 8 * 8); double* ft = tt + B;
        for (size_t i = 0; i < (size_t)B * 6; i++) st[i] = (real)s_[i];
        for (size_t i = 0; i < (size_t)B * 3; i++) ct[i] = (real)c_[i];
        if (o_) for (size_t i = 0; i < (size_t)B * 4; i++) ot[i] = (real)o_[i];
        memcpy(tt, t0, (size_t)B * 8);
        if (toff) memcpy(ft, toff, (size_t)B * 8);
        HIPCHK(h, hipMemcpyAsync(h->d_state, st, (size_t)B * 6 * sizeof(real), kind, h->stream));
        HIPCHK(h, hipMemcpyAsync(h->d_control, ct, (size_t)B * 3 * sizeof(real), kind, h->stream));
        if (o_) HIPCHK(h, hipMemcpyAsync(h->d_other, ot, (size_t)B * 4 * sizeof(real), kind, h->stream));
        else HIPCHK(h, hipMemsetAsync(h->d_other, 0, (size_t)B * 4 * sizeof(real), h->stream));
        HIPCHK(h, hipMemcpyAsync(h->d_t0, tt, (size_t)B * 8, kind, h->stream));
        if (toff) HIPCHK(h, hipMemcpyAsync(h->d_toff, ft, (size_t)B * 8, kind, h->stream));
        else HIPCHK(h, hipMemsetAsync(h->d_toff, 0xFF, (size_t)B * 8, h->stream));      // all-ones bit pattern is a NaN: path-tracking mode
        return PG_OK;
    }
    HIPCHK(h, hipMemcpyAsync(h->d_state, state, (size_t)B * 6 * sizeof(real), kind, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_control, control, (size_t)B * 3 * sizeof(real), kind, h->stream));
    if (other) HIPCHK(h, hipMemcpyAsync(h->d_other, other, (size_t)B * 4 * sizeof(real), kind, h->stream));
    else HIPCHK(h, hipMemsetAsync(h->d_other, 0, (size_t)B * 4 * sizeof(real), h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_t0, t0, (size_t)B * 8, kind, h->stream));
    if (toff) HIPCHK(h, hipMemcpyAsync(h->d_toff, toff, (size_t)B * 8, kind, h->stream));
    else HIPCHK(h, hipMemsetAsync(h->d_toff, 0xFF, (size_t)B * 8, h->stream));      // all-ones bit pattern is a NaN: path-tracking mode
    return PG_OK;
}
int pg_set_inputs(pg_handle* h, int32_t B, const double* state, const double* control, const double* t0, const double* other, const double* toff) {
    return set_inputs(h, B, state, control, t0, other, toff, true);
}
int pg_set_inputs_dev(pg_handle* h, int32_t B, const void* state, const void* control, const double* t0, const void* other, const double* toff) {
    return set_inputs(h, B, state, control, t0, other, toff, false);
}

static int check_ready(pg_handle* h) {
    if (!h) return PG_ERR_INVALID;
    if (h->B <= 0) { h->err = "no inputs installed (call pg_set_inputs first)"; return PG_ERR_STATE; }
    if (!h->d_traj) { h->err = "no trajectory installed (call pg_set_trajectory first)"; return PG_ERR_STATE; }
    if (h->dc.n_traj > 1 && h->traj_idx_B < h->B) { h->err = "a trajectory library is installed but pg_set_trajectory_index does not cover the batch"; return PG_ERR_STATE; }
    if (hipSetDevice(h->cfg.device) != hipSuccess) { h->err = "hipSetDevice failed"; return PG_ERR_HIP; }
    return PG_OK;
}

int pg_compute_time_steps(pg_handle* h) {
    int rc = check_ready(h); if (rc) return rc;
    const int B = h->B;
    hipLaunchKernelGGL(k_time_steps, dim3((B + 63) / 64), dim3(64), 0, h->stream, h->dc, B, h->d_t0, h->d_ts, h->d_dt, h->d_prev_ts);
    LAUNCH_CHECK(h);
    return PG_OK;
}
static int launch_nodes(pg_handle* h, bool with_time_grid);
int pg_compute_linearization_nodes(pg_handle* h) {
    int rc = check_ready(h); if (rc) return rc;
    return launch_nodes(h, false);
}
// compute_time_steps! + compute_linearization_nodes! of pg_step_dev / pg_simulate_dev: the time grid rides in the projection kernel (one launch fewer)
// the pipelined nodes + update_QP launch (k_nodes_linearize) serves the steps of pg_step_dev / pg_simulate_dev when: coupled formulation, some instance is cold (an
// all-warm batch has no recurrence: k_nodes_warm), the batch is large enough for the linearisation to need more than one round of wavefronts (one lane per interval:
// 2048 x 29 / 64 = 928 wavefronts are one round and the two launches are shorter -- 0.310 against 0.315 ms; 2560: 0.386 against 0.317) and small enough for the nodes
// wavefronts to be resident at once (<= 16384 instances = 256 of the 1024 SIMD slots; measured with the one-lane linearisation: 12288: 0.72 against 0.86 ms, 16384: 1.02 against 1.07), and the
// linearisation is not fused into the solve kernel.  With a safety row, its (M, b) -- functions of the measured states only -- are computed BEFORE the launch and the
// launch order is re-filed after it (k_order_hji needs the verdicts the recurrence files).
static int launch_hji_rows_compute(pg_handle* h);
static int launch_hji_order(pg_handle* h);
static bool pipeline_applies(const pg_handle* h) {
    const DevCfg& C = h->dc;
    const bool fuse_wanted = h->fuse == 1;
    return h->pipeline == 1 && C.formulation != PG_DECOUPLED && h->warm_B < h->B && h->B >= h->pipe_min && h->B <= h->pipe_max && C.Ns > 0 && C.Ns < C.N && !fuse_wanted;
}
static int launch_nodes(pg_handle* h, bool with_time_grid) {
    const int B = h->B;
    const bool pipelined = with_time_grid && pipeline_applies(h);
    h->lin_done = false;
    const size_t cap = (size_t)h->cfg.batch_capacity;
    const bool file = h->dc.formulation != PG_DECOUPLED && h->dc.polish;
    int* const order_cnt = file ? h->d_order + cap : (int*)nullptr;            // the two counters of the launch order start from zero: the projection kernel clears them (no memset of its own)
    int* const solve_ctl = h->d_todo + cap;                                    // ... and the control words of this step's solve launches (launch_solve): the to-do counter it counts into
    h->cnt_cleared = !h->sg.capturing;
    if (with_time_grid) hipLaunchKernelGGL(k_project<true>, dim3((B * 64 + 255) / 256), dim3(256), 0, h->stream, h->dc, B, h->d_state, h->d_sep, h->d_t0, h->d_ts, h->d_dt, h->d_prev_ts,
                                           pipelined ? h->d_progress : (int*)nullptr, (B + NODES_IPB - 1) / NODES_IPB + 1, order_cnt, solve_ctl, h->solve_parity);      // (+ 1: the fault word of the pipelined launch)
    else hipLaunchKernelGGL(k_project<false>, dim3((B * 64 + 255) / 256), dim3(256), 0, h->stream, h->dc, B, h->d_state, h->d_sep, (const double*)nullptr, (double*)nullptr, (double*)nullptr,
                            (double*)nullptr, (int*)nullptr, 0, order_cnt, solve_ctl, h->solve_parity);
    LAUNCH_CHECK(h);
    const bool staged = h->dc.n_traj == 1 && h->traj_L <= 2048;
    const size_t traj_lds = staged ? (size_t)2 * h->traj_L * sizeof(real) : 0;
    const dim3 grid((B + 63) / 64), block(64);
    OrderOut F{h->d_status, h->d_iters, h->d_polish, file ? h->d_order : nullptr, h->d_order + cap, h->d_order + cap + 2};
    h->order_B = file ? B : 0;
    if (h->dc.formulation == PG_DECOUPLED) {
        auto kern = staged ? k_nodes_dec<true> : k_nodes_dec<false>;
        hipLaunchKernelGGL(kern, dim3((unsigned)((B + NODES_DEC_IPB - 1) / NODES_DEC_IPB)), block, traj_lds, h->stream, h->dc, B, h->d_state, h->d_control, h->d_toff, h->d_sep, h->d_ts, h->d_dt, h->d_nodes, h->d_naux);      // (NODES_DEC_LPN lanes per instance)
        LAUNCH_CHECK(h);
        const long nn = (long)B * h->dc.NN;
        hipLaunchKernelGGL(k_nodes_angles, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, h->stream, h->dc, B, h->d_naux, h->d_nodes, (const int*)nullptr);
    } else {
        if (h->warm_B >= B) {           // every instance takes the warm branch: lane = (instance, node)
            auto kern = staged ? k_nodes_warm<true> : k_nodes_warm<false>;
            const long nth = (long)B * h->dc.NN;
            hipLaunchKernelGGL(kern, dim3((unsigned)((nth + 255) / 256)), dim3(256), traj_lds, h->stream, h->dc, B, h->d_state, h->d_control, h->d_sep, h->d_ts, h->d_prev_ts, h->d_solx,
                               h->d_nodes, F);
        } else if (pipelined) {
            { int rc = launch_hji_rows_compute(h); if (rc) return rc; }      // (M, b) of the safety row: read by the lanes that linearise interval 0
            const int lpi = h->lin_lpi, ipb = 64 / lpi;
            const int nbn = (B + NODES_IPB - 1) / NODES_IPB, nbt = (B + ipb - 1) / ipb;
            const size_t lds = traj_lds > 64 * 20 * sizeof(real) ? traj_lds : 64 * 20 * sizeof(real);
            auto kern = lpi == 1 ? (staged ? k_nodes_linearize<true, 1> : k_nodes_linearize<false, 1>) : (staged ? k_nodes_linearize<true, 2> : k_nodes_linearize<false, 2>);
#ifdef PG_F32
            if (lpi == 1 && B >= 6144) kern = staged ? k_nodes_linearize<true, 1, 2> : k_nodes_linearize<false, 1, 2>;      // (two waves per SIMD once the linearisation is more than one round of them)
#endif
            int nzf = (1024 - nbn + nbt - 1) / nbt;               // short-horizon intervals that go first: one wavefront for every SIMD the recurrence leaves free
            if (h->pipe_first > 0) nzf = h->pipe_first;
            if (nzf > h->dc.Ns) nzf = h->dc.Ns;
            if (nzf < 1) nzf = 1;
            // nodes after which the recurrence publishes its progress (each publication is a device-scope release, i.e. an L2 write-back): every third node of the
            // short horizon, every fifth of the long one (N = 30, Ns = 10: nodes 3, 6, 9, 14, 19, 24); the end of the recurrence always publishes
            unsigned long long pub = 0ull;
            for (int i = 1; i < h->dc.N - 2 && i < 63; i++)
                if (i <= h->dc.Ns ? i % h->pipe_pub_short == 0 : (i - h->dc.Ns) % h->pipe_pub_long == h->pipe_pub_long - 1) pub |= 1ull << i;
            if (h->pipe_fault) pub = 1ull << 63;                  // test hook: nothing is ever published (tests/test_gpu_api_contract.py)
            hipLaunchKernelGGL(kern, dim3((unsigned)(nbn + nbt * h->dc.N)), block, lds, h->stream, h->dc, B, nbn, nzf, pub, h->d_state, h->d_control, h->d_toff, h->d_solved, h->d_sep, h->d_ts,
                               h->d_dt, h->d_prev_ts, h->d_solx, h->d_nodes, F, h->d_naux, h->d_progress, h->d_Mb, h->d_qp, h->d_progress + nbn, h->d_progress + cap / NODES_IPB + 7);
            LAUNCH_CHECK(h);
            // repair, queued unconditionally and predicated on the device: if any waiting wavefront of the launch above gave up (its fault word, zeroed by the
            // projection kernel of the step), the deferred angles and update_QP! of the WHOLE batch run launch per phase -- the same kernels on the same nodes, so the QP
            // data are the ones the pipeline would have written.  When nothing gave up the two launches return at once (~3 us together).
            {
                const int* flt = h->d_progress + nbn;
                const long nn = (long)B * h->dc.NN;
                (void)nn;
                hipLaunchKernelGGL(k_nodes_angles, dim3(64), dim3(256), 0, h->stream, h->dc, B, h->d_naux, h->d_nodes, flt);               // (small grids, striding over the work:
                const long nz = (long)B * h->dc.Ns * lpi, nr = (long)B * (h->dc.N - h->dc.Ns) * lpi;                                           //  an empty launch is cheap)
                const int nbz = (int)((nz + 63) / 64), nbr = (int)((nr + 63) / 64);
                auto ksplit = lpi == 1 ? k_linearize_split<1> : k_linearize_split<2>;
                hipLaunchKernelGGL(ksplit, dim3(512), dim3(64), 0, h->stream, h->dc, B, nbz, h->d_nodes, h->d_dt, h->d_Mb, h->d_qp, flt, nbz + nbr);
            }
            h->lin_done = true; h->stat_pipelined++;
        } else {
            auto kern = staged ? k_nodes<true> : k_nodes<false>;
            hipLaunchKernelGGL(kern, dim3((B + NODES_IPB - 1) / NODES_IPB), block, traj_lds, h->stream, h->dc, B, h->d_state, h->d_control, h->d_toff, h->d_solved, h->d_sep, h->d_ts,
                               h->d_dt, h->d_prev_ts, h->d_solx, h->d_nodes, F, h->d_naux);
            LAUNCH_CHECK(h);
            const long nn = (long)B * h->dc.NN;
            hipLaunchKernelGGL(k_nodes_angles, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, h->stream, h->dc, B, h->d_naux, h->d_nodes);
        }
    }
    LAUNCH_CHECK(h);
    return PG_OK;
}
static int launch_hji_lookup(pg_handle* h, int B, const real* x7_dev, real* out8_dev) {
    dim3 grid((unsigned)(((size_t)B * 16 + 255) / 256));
    if (h->hv.cdims == 7) hipLaunchKernelGGL(k_hji_lookup<7>, grid, dim3(256), 0, h->stream, h->hv, B, x7_dev, out8_dev);
    else if (h->hv.cdims == 5) hipLaunchKernelGGL(k_hji_lookup<5>, grid, dim3(256), 0, h->stream, h->hv, B, x7_dev, out8_dev);
    else hipLaunchKernelGGL(k_hji_lookup<3>, grid, dim3(256), 0, h->stream, h->hv, B, x7_dev, out8_dev);
    LAUNCH_CHECK(h);
    return PG_OK;
}
// the safety row of update_QP! (coupled_lat_long.jl:345-346): relative state, value / gradient look-up, (M, b) per instance
// (two halves: the rows themselves read the measured states only and may run before the nodes -- the pipelined launch needs (M, b) when it linearises interval 0 --,
// the re-filed launch order needs the verdicts the nodes kernel filed)
static int launch_hji_rows_compute(pg_handle* h) {
    if (!h->has_hji) return PG_OK;
    const int B = h->B;
    hipLaunchKernelGGL(k_hji_relstate, dim3((B + 255) / 256), dim3(256), 0, h->stream, B, h->d_state, h->d_other, h->d_x7);
    LAUNCH_CHECK(h);
    int rc = launch_hji_lookup(h, B, h->d_x7, h->d_vg8); if (rc) return rc;
    hipLaunchKernelGGL(k_hji_constraint, dim3((B + 63) / 64), dim3(64), 0, h->stream, h->dc, B, h->d_x7, h->d_vg8, h->d_control, h->d_Mb);
    LAUNCH_CHECK(h);
    return PG_OK;
}
static int launch_hji_order(pg_handle* h) {
    if (!h->has_hji) return PG_OK;
    const int B = h->B;
    if (h->dc.polish && h->order_B == B) {              // launch order again, now that the safety rows are known (k_order_hji)
        const size_t cap = (size_t)h->cfg.batch_capacity;
        OrderOut F{h->d_status, h->d_iters, h->d_polish, h->d_order, h->d_order + cap, h->d_order + cap + 2};
        HIPCHK(h, hipMemsetAsync(h->d_order + cap, 0, 2 * sizeof(int), h->stream));
        hipLaunchKernelGGL(k_order_hji, dim3((B + 255) / 256), dim3(256), 0, h->stream, h->dc, B, h->d_control, h->d_Mb, F);
        LAUNCH_CHECK(h);
    }
    return PG_OK;
}
static int launch_hji_rows(pg_handle* h) {
    int rc = launch_hji_rows_compute(h); if (rc) return rc;
    return launch_hji_order(h);
}
#define PG_LIN_PAIR_MAX 1024
static int launch_linearize(pg_handle* h, int n) {
    // two lanes per (instance, interval) with four tangent directions each; small batches -- a handful of wavefronts whose duration is the latency of one lane --
    // spread the eight directions over four or eight lanes instead (same arithmetic per direction)
    // Large batches (several rounds of wavefronts): ONE lane per interval with all eight directions (k_linearize_split<1>) -- the fewest instructions per interval.
    // While every lane pair is resident at once (n <= 1024: 2 x 29 x 1024 / 64 = 928 wavefronts on 1024 SIMDs) the pair is the shorter chain.
    const int g_env = h->lin_groups;       // (-DPG_DIAG builds: option "diag_lin_groups" forces the lane arrangement; 0 otherwise)
#ifdef PG_F32
    const int G_small = 2;   // (fp32: the instantiations differ at rounding level -- the compiler contracts them differently --, so only the two classes n <= 1024 / above exist:
                             //  shards and whole batch agree bit for bit when they fall in the same class)
#else
    const int G_small = n <= 256 ? 8 : (n <= 512 ? 4 : 2);
#endif
    // fp64: bit-identical across K (tests/test_gpu_multiprocess.py steps 1024 = 2 x 512, 1001 = 501 + 500, 400 = 2 x 200; tests/test_gpu_full_size.py 4096 against 4 x 1024)
    const int G = (g_env == 1 || g_env == 2 || g_env == 4 || g_env == 8) ? g_env : ((n <= PG_LIN_PAIR_MAX || h->lin_lpi == 2) ? G_small : 1);
    if (G <= 2 && h->dc.Ns > 0 && h->dc.Ns < h->dc.N) {     // short-horizon intervals without the two uf directions (k_linearize_split)
        const long nz = (long)n * h->dc.Ns * G, nr = (long)n * (h->dc.N - h->dc.Ns) * G;
        const int nbz = (int)((nz + 63) / 64), nbr = (int)((nr + 63) / 64);
        auto ksplit = G == 1 ? k_linearize_split<1> : k_linearize_split<2>;
        hipLaunchKernelGGL(ksplit, dim3((unsigned)(nbz + nbr)), dim3(64), 0, h->stream, h->dc, n, nbz, h->d_nodes, h->d_dt, h->d_Mb, h->d_qp, (const int*)nullptr, 0);
        LAUNCH_CHECK(h);
        return PG_OK;
    }
    const long nl = (long)n * h->dc.N * G;
    const dim3 grid((unsigned)((nl + 63) / 64)), block(64);
    if (G == 1) hipLaunchKernelGGL(k_linearize<8>, grid, block, 0, h->stream, h->dc, n, h->d_nodes, h->d_dt, h->d_Mb, h->d_qp);
    else if (G == 8) hipLaunchKernelGGL(k_linearize<1>, grid, block, 0, h->stream, h->dc, n, h->d_nodes, h->d_dt, h->d_Mb, h->d_qp);
    else if (G == 4) hipLaunchKernelGGL(k_linearize<2>, grid, block, 0, h->stream, h->dc, n, h->d_nodes, h->d_dt, h->d_Mb, h->d_qp);
    else hipLaunchKernelGGL(k_linearize<4>, grid, block, 0, h->stream, h->dc, n, h->d_nodes, h->d_dt, h->d_Mb, h->d_qp);
    LAUNCH_CHECK(h);
    return PG_OK;
}
int pg_update_qp(pg_handle* h) {
    int rc = check_ready(h); if (rc) return rc;
    const int B = h->B; const DevCfg& C = h->dc;
    if (C.formulation == PG_DECOUPLED) {
        long nt = (long)B * C.N;
        // (a handle whose solver is k_solve_lat needs the packed stage records only: the embedded block is built when pg_get_qp asks for it)
        const int embed = (h->solve_lat && h->lat_pack_only) ? 0 : 1;
        hipLaunchKernelGGL(k_qp_dec, dim3((unsigned)((nt + 127) / 128)), dim3(128), 0, h->stream, h->dc, B, h->d_nodes, h->d_dt, h->d_qp, embed);
        LAUNCH_CHECK(h);
        h->qp_embedded = embed != 0; h->qp_stale = false;
        return PG_OK;
    }
    if ((rc = launch_hji_rows(h))) return rc;
    return launch_linearize(h, B);
}
// k_solve over `n` instances on stream `st`: the whole batch in index order (order == nullptr) or the sub-range order[0..n) of the launch order
static int launch_solve(pg_handle* h, hipStream_t st, const int* order, int n, unsigned long long* lat_prof = nullptr) {
    SolveOut O{h->d_solx, h->d_sigma, h->d_u, h->d_status, h->d_iters, h->d_active, h->d_mu, h->d_solved, h->d_polish, h->d_lam, order, h->d_wfail, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
#ifdef PG_EXPERIMENTAL_SOLVE4
    if (h->solve_quad) { hipLaunchKernelGGL((k_solve4<2, false>), dim3((h->B + 3) / 4), dim3(64), h->solve4_lds, st, h->dc, h->B, h->d_qp, h->d_nodes, h->d_ws4, O, (unsigned long long*)nullptr); LAUNCH_CHECK(h); return PG_OK; }
#endif
    if (h->solve_lat) {      // lateral formulation: four instances per wavefront, always the whole batch in index order
        const dim3 grid((unsigned)((h->B + 3) / 4));
#ifdef PG_DIAG
        if (h->dc.dbg_poison) {      // (every byte 0xFF: NaN in both precisions)
            const size_t cap = (size_t)h->cfg.batch_capacity;
            if (h->d_lat_ws) HIPCHK(h, hipMemsetAsync(h->d_lat_ws, 0xFF, lat_ws_bytes(cap), st));
            if (h->d_lat_aux) HIPCHK(h, hipMemsetAsync(h->d_lat_aux, 0xFF, (cap + 1) * 64 * LAT_AUX * sizeof(real), st));
            if (h->d_lat_spc) HIPCHK(h, hipMemsetAsync(h->d_lat_spc, 0xFF, lat_spc_bytes((int)cap, h->dc.N), st));
        }
#endif
        const int slots = (h->dc.N + 15) / 16;
#define PG_LAT_LAUNCH(NS, W, M) hipLaunchKernelGGL((k_solve_lat<NS, W, M>), grid, dim3(64), h->lat_lds, st, h->dc, h->B, h->d_qp, h->d_nodes, O, lat_prof)
#define PG_LAT_LAUNCH_ANY() do { \
        if (h->dc.walls) { if (h->lat_mem) PG_LAT_LAUNCH(1, true, true); else PG_LAT_LAUNCH(1, true, false); } \
        else { if (h->lat_mem) PG_LAT_LAUNCH(1, false, true); else PG_LAT_LAUNCH(1, false, false); } } while (0)
        // A batch in which every instance carries a previous solution (a closed loop after its first step) is solved in TWO launches: the warm attempts, then -- over the
        // list the first launch leaves -- the cold solves of what they did not serve, packed four per wavefront again (see k_solve_lat).  Option "lat_split" = 0: one launch.
        const bool two = h->split_lat && h->dc.polish && h->dc.warm_polish && h->dc.lat_warm_rounds > 0 && h->warm_B >= h->B && !lat_prof && !h->sg.capturing;
        if (two) {
            const size_t cap = (size_t)h->cfg.batch_capacity;
            HIPCHK(h, hipMemsetAsync(h->d_todo + cap, 0, sizeof(int), st)); h->stat_lat_two++;
            O.todo = h->d_todo; O.n_todo = h->d_todo + cap;
            PG_LAT_LAUNCH_ANY();
            LAUNCH_CHECK(h);
            O.todo = nullptr; O.n_todo = nullptr; O.list = h->d_todo; O.n_list = h->d_todo + cap;
            // (round 6) a list of up to `lat_single_max` instances is solved ONE instance per wavefront (see "small batches" below: 871 unserved instances of the benchmark
            // batch 1.5 instead of 2.1 ms); the host does not know the length, so both arrangements are queued and the device word picks one (an idle launch: ~4 us)
            if (h->lat_single_max > 0 && h->dc.N > 16) {
                const unsigned nb1 = (unsigned)(h->B < h->lat_single_max ? h->B : h->lat_single_max);
                const size_t lds1 = lat_lds_doubles(h->dc.N, 1) * sizeof(real);
                O.list_lo = 0; O.list_hi = h->lat_single_max;
                if (h->dc.walls) hipLaunchKernelGGL((k_solve_lat<1, true, false, 64, 0>), dim3(nb1), dim3(64), lds1, st, h->dc, h->B, h->d_qp, h->d_nodes, O, lat_prof);
                else hipLaunchKernelGGL((k_solve_lat<1, false, false, 64, 0>), dim3(nb1), dim3(64), lds1, st, h->dc, h->B, h->d_qp, h->d_nodes, O, lat_prof);
                LAUNCH_CHECK(h);
                O.list_lo = h->lat_single_max + 1; O.list_hi = 0;
            }
        }
        // Straggler hand-over (round 6, see k_solve_lat): a batch that starts cold interior points runs as TWO launches -- the first stops at a trip boundary once at most
        // `lat_hand_target` instances of the batch are unfinished and files them, the second resumes those, ONE instance per wavefront (option "lat_handover" = 0: one launch
        // as in round 5.  Measured and removed: resuming four per wavefront again, i.e. compaction only -- 3.34 against 3.25 ms for the single launch).  Only where the row
        // state lives in the workspace (N > 32, or the wall rows), and not behind the warm attempts' own two launches.
        // Small batches (round 6): with at most one wavefront per SIMD to fill anyway, every instance gets a wavefront of its own from the start -- the mapping of the resuming
        // launch (lane = stage in the stage-parallel passes, one slot visit instead of four, the row state in registers): a trip costs 57 us instead of 84.  Not behind warm
        // attempts (their launches are a round or two of the polish: four per wavefront is the cheaper shape there).
        if (!two && !h->sg.capturing && h->warm_B < h->B && h->B <= h->lat_single_max && h->dc.N > 16) {
            const size_t lds1 = lat_lds_doubles(h->dc.N, 1) * sizeof(real);
            if (h->dc.walls) hipLaunchKernelGGL((k_solve_lat<1, true, false, 64, 0>), dim3((unsigned)h->B), dim3(64), lds1, st, h->dc, h->B, h->d_qp, h->d_nodes, O, lat_prof);
            else hipLaunchKernelGGL((k_solve_lat<1, false, false, 64, 0>), dim3((unsigned)h->B), dim3(64), lds1, st, h->dc, h->B, h->d_qp, h->d_nodes, O, lat_prof);
            LAUNCH_CHECK(h); h->stat_lat_single++;
            return PG_OK;
        }
        const bool hand = h->lat_handover != 0 && h->lat_mem && !two && !h->sg.capturing && h->B >= h->lat_hand_batch ;
        if (hand) {
            const size_t cap = (size_t)h->cfg.batch_capacity;
            int* const ctl = h->d_todo + cap;
            HIPCHK(h, hipMemsetAsync(ctl, 0, 3 * sizeof(int), st)); h->stat_lat_hand++;      // [0] listed from the front, [1] finished instances, [2] listed from the back
            O.todo = h->d_todo; O.n_todo = ctl; O.hand_mode = 1;
            // When to stop the first launch.  Default: after a fixed number of trips (16 with the wall rows, 11 without: mean interior-point iterations + 2..3 on the benchmark
            // batches) -- a rule that depends on the data only, so that the same call gives the same bits.  Option "lat_hand_target" > 0 stops when that few instances of the
            // batch are unfinished instead (counted on the device: adapts to the batch -- vail + walls 2.09 against 2.19 ms -- but WHEN a wavefront sees the count is a matter
            // of timing, and an instance resumed one trip earlier or later ends 1e-8 away: two verified KKT points of the same QP, not the same bits).
            O.hand_cap = h->lat_hand_cap > 0 ? h->lat_hand_cap : ((h->lat_hand_target > 0 || h->lat_hand_work > 0) ? 0 : (h->dc.walls ? 16 : 11)); O.hand_target = h->lat_hand_target; O.hand_min = h->lat_hand_min; O.hand_done = ctl + 1; O.hand_work = h->lat_hand_work; O.hand_w0 = h->lat_hand_w0;
            O.hand_r = h->d_hand_r; O.hand_i = h->d_hand_i;
            if (h->dc.walls) hipLaunchKernelGGL((k_solve_lat<1, true, true, 16, 1>), grid, dim3(64), h->lat_lds, st, h->dc, h->B, h->d_qp, h->d_nodes, O, lat_prof);
            else hipLaunchKernelGGL((k_solve_lat<1, false, true, 16, 1>), grid, dim3(64), h->lat_lds, st, h->dc, h->B, h->d_qp, h->d_nodes, O, lat_prof);
            LAUNCH_CHECK(h);
            O.todo = nullptr; O.n_todo = nullptr; O.list = h->d_todo; O.n_list = ctl; O.hand_mode = 2;
            const size_t lds1 = lat_lds_doubles(h->dc.N, 1) * sizeof(real);          // one wavefront per listed instance; blocks beyond the list return at once
            if (h->dc.walls) hipLaunchKernelGGL((k_solve_lat<1, true, false, 64, 2>), dim3((unsigned)h->B), dim3(64), lds1, st, h->dc, h->B, h->d_qp, h->d_nodes, O, lat_prof);
            else hipLaunchKernelGGL((k_solve_lat<1, false, false, 64, 2>), dim3((unsigned)h->B), dim3(64), lds1, st, h->dc, h->B, h->d_qp, h->d_nodes, O, lat_prof);
            LAUNCH_CHECK(h);
            return PG_OK;
        }
        PG_LAT_LAUNCH_ANY();
#undef PG_LAT_LAUNCH_ANY
#undef PG_LAT_LAUNCH
        LAUNCH_CHECK(h);
        return PG_OK;
    }
    if (h->u_direct) { O.u_out2 = h->u_direct; h->u_written = true; }      // (every k_solve instantiation below writes the caller's array itself)
    if (h->solve_ring) hipLaunchKernelGGL((k_solve<false, true, false>), dim3(n), dim3(64), h->solve_lds, st, h->dc, h->B, h->d_qp, h->d_nodes, O, (unsigned long long*)nullptr, h->d_dt, h->d_Mb);
    else if (h->split_solve && h->dc.polish && (h->dc.cold_guess > 0 || h->dc.warm_polish) && !h->has_hji && !h->sg.capturing) {
        // Two launches: the rounds-only instantiation (no scratch: the interior point's state and code are not in it) serves the instances an active-set attempt
        // verifies -- all of them on the tracking batches --; what it leaves (SolveOut::todo) goes through the full kernel in list mode, its blocks returning at once
        // when the list is empty.  Not with a safety row installed: there the instances whose row is violated NEED the interior point (9-17 % of config 3), and in one
        // kernel they start first (launch order) instead of after everyone else.  For the same reason a launch of this handle that has left something for the interior
        // point changes the NEXT launch: the full kernel then takes the whole batch in its launch order and the rounds-only kernel returns at once -- an instance that needs
        // the interior point runs it right behind its rounds, under the rest of the batch, instead of after it (`vail`, two such instances of 4096: 1.38 ms per cold step
        // split, 1.07 ms in one kernel).  The full kernel counts its interior-point instances the same way, so the split returns when they are gone.
        // The decision is taken ON THE DEVICE from the previous launch's count (SolveOut::mode), i.e. from stream-ordered state only: the same sequence of calls gives the
        // same launches whatever the host's timing (round 4 read a pinned copy of the count that an asynchronous copy filled "whenever it arrived").  Two counters are used
        // alternately -- this launch counts into one while the other still holds the previous launch's count --; the projection kernel of the step zeroes the one about to be
        // used (a solve without a nodes phase in front -- replayed QPs -- zeroes it here).
        const size_t cap = (size_t)h->cfg.batch_capacity;
        int* const ctl = h->d_todo + cap; int* const cnt = ctl + h->solve_parity; const int* const prev = ctl + (h->solve_parity ^ 1);
        if (!h->cnt_cleared) HIPCHK(h, hipMemsetAsync(cnt, 0, sizeof(int), st));
        h->cnt_cleared = false; h->solve_parity ^= 1; h->stat_split++;
        O.todo = h->d_todo; O.n_todo = cnt; O.mode = prev; O.n_whole = ctl + 5;
        hipLaunchKernelGGL((k_solve<false, false, false, false>), dim3(n), dim3(64), h->solve_lds, st, h->dc, h->B, h->d_qp, h->d_nodes, O, (unsigned long long*)nullptr, h->d_dt, h->d_Mb);
        LAUNCH_CHECK(h);
        SolveOut O2 = O; O2.todo = nullptr; O2.list = h->d_todo; O2.n_list = cnt;      // (order_in stays: the whole-batch mode uses the launch order)
        hipLaunchKernelGGL((k_solve<false, false, false, true>), dim3(n), dim3(64), h->solve_lds, st, h->dc, h->B, h->d_qp, h->d_nodes, O2, (unsigned long long*)nullptr, h->d_dt, h->d_Mb);
    }
    else {
        h->stat_single++;
        hipLaunchKernelGGL((k_solve<false, false, false>), dim3(n), dim3(64), h->solve_lds, st, h->dc, h->B, h->d_qp, h->d_nodes, O, (unsigned long long*)nullptr, h->d_dt, h->d_Mb);
    }
    LAUNCH_CHECK(h);
    return PG_OK;
}
int pg_solve(pg_handle* h) {
    int rc = check_ready(h); if (rc) return rc;
    if (h->qp_stale) { h->err = "pg_solve: the lateral solver was switched (pg_set_option \"lateral_solver\") after the last update_QP!: call pg_update_qp first"; return PG_ERR_STATE; }
    // launch order: the one the nodes kernels of this step filed (likely slow instances first)
    const bool use_order = h->dc.polish && h->order_B == h->B;
    if ((rc = launch_solve(h, h->stream, use_order ? h->d_order : nullptr, h->B))) return rc;
    if (h->B > h->warm_B) h->warm_B = h->B;           // model_predictive_control.jl:76: solved = true for every instance of the batch
    return PG_OK;
}
// update_QP! + solve! of one step.  For the coupled formulation with N <= 32 both run in ONE kernel (k_solve<.., FUSE = true>): the wavefront that solves an instance
// first linearises it (two lanes per interval = 60 of its 64 lanes), writes the QP data (still readable through pg_get_qp) and goes on to the solve.  k_solve ends
// with its slowest wave -- an instance that needs the interior point takes 0.5 ms whatever the batch size -- and as a separate kernel it leaves part of the machine
// idle for the last third of its run; fused, the SIMDs that are done with their quick instances linearise and solve the next ones meanwhile.  Same device
// functions, same per-instance arithmetic: results are bit-identical to the two-kernel sequence.  MEASURED (B = 4096, MI355X): 1.23 vs 1.30 ms per cold step on
// skidpadoval, but 1.52 vs 1.40 on vail and 1.13 vs 1.05 on EastPaddock -- the linearisation runs ~15 % slower inside the big kernel (the waves of a CU are then
// spread over 85 KB of code instead of sharing one loop in the 64 KB instruction cache), and where no straggler tail exists there is nothing to win.  Closed
// loops were 5-8 % faster fused as long as a few instances per step fell back to the interior point; with those stragglers gone (the polish rules of k_solve)
// the two-kernel sequence wins there too (0.91 vs 0.99, 1.06 vs 1.11, 0.87 vs 0.98 ms per closed-loop step).  Hence OFF by default (pg_set_fusion:
// 0 never, 1 always, 2 for all-warm batches of >= 1024 instances).
static int update_and_solve(pg_handle* h, hipEvent_t after_update) {
    int rc;
    const bool want = h->fuse == 1 || (h->fuse == 2 && h->warm_B >= h->B && h->B >= 1024);
    const bool fused = want && h->dc.formulation != PG_DECOUPLED && !h->solve_ring && 2 * h->dc.N <= 64 && !h->solve_quad;
    if (h->lin_done) {                   // the QP data of this step are already there (k_nodes_linearize)
        h->lin_done = false;
        if ((rc = launch_hji_order(h))) return rc;
        if (after_update) HIPCHK(h, hipEventRecord(after_update, h->stream));
        return pg_solve(h);
    }
    if (!fused) {
        if ((rc = pg_update_qp(h))) return rc;
        if (after_update) HIPCHK(h, hipEventRecord(after_update, h->stream));
        return pg_solve(h);
    }
    if ((rc = launch_hji_rows(h))) return rc;
    if (after_update) HIPCHK(h, hipEventRecord(after_update, h->stream));
    const bool use_order = h->dc.polish && h->order_B == h->B;
    SolveOut O{h->d_solx, h->d_sigma, h->d_u, h->d_status, h->d_iters, h->d_active, h->d_mu, h->d_solved, h->d_polish, h->d_lam, use_order ? h->d_order : nullptr, h->d_wfail, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (h->u_direct) { O.u_out2 = h->u_direct; h->u_written = true; }
    hipLaunchKernelGGL((k_solve<false, false, true>), dim3(h->B), dim3(64), h->solve_lds, h->stream, h->dc, h->B, h->d_qp, h->d_nodes, O, (unsigned long long*)nullptr, h->d_dt, h->d_Mb);
    LAUNCH_CHECK(h);
    if (h->B > h->warm_B) h->warm_B = h->B;
    return PG_OK;
}
#ifdef PG_DIAG
// debug (-DPG_DIAG library only; not part of the public header, not in the shipped libraries -- nor are the PROF instantiations of k_solve it launches): per-phase shader-clock
// cycles of one solve launch, out [B][6] = (stage assembly+step, sync, matrix pass, vector passes, forward passes, prologue); diagnostic build of the kernel, never timed
int pg_debug_solve_cycles(pg_handle* h, unsigned long long* out) {
    int rc = check_ready(h); if (rc) return rc;
    unsigned long long* d = nullptr;
    HIPCHK(h, hipMalloc((void**)&d, ((size_t)h->B * 9 + 1024) * 8));
    HIPCHK(h, hipMemset(d, 0, ((size_t)h->B * 9 + 1024) * 8));
    SolveOut O{h->d_solx, h->d_sigma, h->d_u, h->d_status, h->d_iters, h->d_active, h->d_mu, h->d_solved, h->d_polish, h->d_lam,
               (h->dc.polish && h->order_B == h->B) ? h->d_order : nullptr, h->d_wfail, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};            // (the launch order pg_solve would use: the timeline is the product's)
    if (h->solve_lat) { if ((rc = launch_solve(h, h->stream, nullptr, h->B, d))) { (void)hipFree(d); return rc; } } else
#ifdef PG_EXPERIMENTAL_SOLVE4
    if (h->solve_quad) hipLaunchKernelGGL((k_solve4<2, true>), dim3((h->B + 3) / 4), dim3(64), h->solve4_lds, h->stream, h->dc, h->B, h->d_qp, h->d_nodes, h->d_ws4, O, d);
    else
#endif
    if (h->solve_ring) hipLaunchKernelGGL((k_solve<true, true, false>), dim3(h->B), dim3(64), h->solve_lds, h->stream, h->dc, h->B, h->d_qp, h->d_nodes, O, d, h->d_dt, h->d_Mb);
    else if (h->debug_timeline) {      // the product's kernel (the rounds-only instantiation of the split launch): timeline only
        int* const cnt = h->d_todo + (size_t)h->cfg.batch_capacity + 4;
        HIPCHK(h, hipMemsetAsync(cnt, 0, sizeof(int), h->stream));
        O.todo = h->d_todo; O.n_todo = cnt;
        hipLaunchKernelGGL((k_solve<false, false, false, false>), dim3(h->B), dim3(64), h->solve_lds, h->stream, h->dc, h->B, h->d_qp, h->d_nodes, O, d, h->d_dt, h->d_Mb);
    }
    else hipLaunchKernelGGL((k_solve<true, false, false>), dim3(h->B), dim3(64), h->solve_lds, h->stream, h->dc, h->B, h->d_qp, h->d_nodes, O, d, h->d_dt, h->d_Mb);
    LAUNCH_CHECK(h);
    HIPCHK(h, hipMemcpy(out, d, ((size_t)h->B * 9 + 1024) * 8, hipMemcpyDeviceToHost));      // out: [B][6] cycles + 1024-double trace of the "diag_instance" instance + [B][3] timeline (k_solve: entry, exit on the 100 MHz wall clock, HW_ID | XCC_ID << 32)
    (void)hipFree(d);
    return PG_OK;
}
#endif
#ifdef PG_TIMELINE
// debug (timeline build only, not part of the public header): per block of the last pipelined launch (entry, end of the wait, exit on the 100 MHz wall clock; interval, 1000 = a nodes block)
extern "C" int pg_debug_pipeline_timeline(pg_handle* h, unsigned long long* out, int n_blocks) {
    int rc = check_ready(h); if (rc) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpyFromSymbol(out, HIP_SYMBOL(g_nl_timeline), (size_t)(n_blocks < 8192 ? n_blocks : 8192) * 8 * sizeof(unsigned long long)));
    return PG_OK;
}
#endif
int pg_get_next_control_dev(pg_handle* h, void* u_out_dev) {
    int rc = check_ready(h); if (rc) return rc;
    if (u_out_dev) HIPCHK(h, hipMemcpyAsync(u_out_dev, h->d_u, (size_t)h->B * 3 * sizeof(real), hipMemcpyDeviceToDevice, h->stream));
    return PG_OK;
}
int pg_get_next_control(pg_handle* h, double* u_out) {
    int rc = check_ready(h); if (rc) return rc;
    REQUIRE(h, u_out, "u_out is null");
    return down(h, u_out, h->d_u, (size_t)h->B * 3);
}
// ros_integration.jl:114-124: the HJI fallback policy takes the wheel when the value function says the situation is unsafe
static int launch_hji_policy(pg_handle* h, int use_policy) {
    const int B = h->B;
    if (h->dc.formulation != PG_COUPLED) { h->err = "the HJI policy belongs to the coupled controller (ros_integration.jl:56,114)"; return PG_ERR_STATE; }
    if (!h->has_hji) {                      // no grid: V = +Inf everywhere, the MPC control always wins
        HIPCHK(h, hipMemcpyAsync(h->d_pol_u, h->d_u, (size_t)B * 3 * sizeof(real), hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(h, hipMemsetAsync(h->d_pol_src, 0, (size_t)B * sizeof(int), h->stream));
        HIPCHK(h, hipMemsetAsync(h->d_pol_u2, 0, (size_t)B * 2 * sizeof(real), h->stream));
        return PG_OK;
    }
    hipLaunchKernelGGL(k_hji_policy, dim3((B + 63) / 64), dim3(64), 0, h->stream, h->dc, B, use_policy, h->d_x7, h->d_vg8, h->d_toff, h->d_u, h->d_pol_u2, h->d_pol_u, h->d_pol_src);
    LAUNCH_CHECK(h);
    return PG_OK;
}
int pg_get_next_control_hji_dev(pg_handle* h, int32_t use_hji_policy, void* u_out_dev, int32_t* source_dev) {
    int rc = check_ready(h); if (rc) return rc;
    if ((rc = launch_hji_policy(h, use_hji_policy))) return rc;
    if (u_out_dev) HIPCHK(h, hipMemcpyAsync(u_out_dev, h->d_pol_u, (size_t)h->B * 3 * sizeof(real), hipMemcpyDeviceToDevice, h->stream));
    if (source_dev) HIPCHK(h, hipMemcpyAsync(source_dev, h->d_pol_src, (size_t)h->B * sizeof(int), hipMemcpyDeviceToDevice, h->stream));
    return PG_OK;
}
int pg_get_next_control_hji(pg_handle* h, int32_t use_hji_policy, double* u_out, int32_t* source, double* u2_policy) {
    int rc = check_ready(h); if (rc) return rc;
    REQUIRE(h, u_out, "u_out is null");
    if ((rc = launch_hji_policy(h, use_hji_policy))) return rc;
    if ((rc = down(h, u_out, h->d_pol_u, (size_t)h->B * 3)) || (rc = down_raw(h, source, h->d_pol_src, (size_t)h->B * sizeof(int))) ||
        (rc = down(h, u2_policy, h->d_pol_u2, (size_t)h->B * 2))) return rc;
    return PG_OK;
}
int pg_step_dev(pg_handle* h, void* u_out_dev) {
    int rc = check_ready(h); if (rc) return rc;
    const bool ev = !h->sg.capturing && h->phase_timing != 0;     // (a step that is being captured into a graph carries no timing events; option "phase_timing" = 0: none at all)
    if (ev) HIPCHK(h, hipEventRecord(h->ev[0], h->stream));
    if ((rc = launch_nodes(h, true))) return rc;
    if (ev) HIPCHK(h, hipEventRecord(h->ev[1], h->stream));
    h->u_direct = (real*)u_out_dev; h->u_written = false;
    rc = update_and_solve(h, ev ? h->ev[2] : nullptr);       // (chunked: ev[2] marks the end of the LAST update_QP chunk; earlier solve chunks run under it)
    h->u_direct = nullptr;
    if (rc) return rc;
    if (!h->u_written && (rc = pg_get_next_control_dev(h, u_out_dev))) return rc;      // (k_solve writes the caller's array itself; the lateral kernel's controls are copied)
    if (ev) HIPCHK(h, hipEventRecord(h->ev[3], h->stream));
    h->timing_valid = ev;
    return PG_OK;
}
int pg_simulate_dev(pg_handle* h, int32_t steps, double dt, void* state_hist_dev_, void* control_hist_dev_) {
    int rc = check_ready(h); if (rc) return rc;
    REQUIRE(h, steps >= 1 && dt > 0.0, "pg_simulate_dev: steps >= 1 and dt > 0 required");
    const int B = h->B;
    real* state_hist_dev = (real*)state_hist_dev_; real* control_hist_dev = (real*)control_hist_dev_;
    // the loop's clock (:87): t takes the elements of 0:dt:trajectory.t[end], here shifted by each instance's start time.  A call continues the clock of the previous one
    // (same dt, same path end, no pg_set_inputs in between); otherwise it restarts from the times the inputs carry
    if (h->sim_idx == 0 || h->sim_dt != dt || h->sim_tend != h->traj_t_end) {
        HIPCHK(h, hipMemcpyAsync(h->d_tstart, h->d_t0, (size_t)B * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        h->sim_clk = jl_colon(0.0, dt, h->traj_t_end); h->sim_idx = 1; h->sim_dt = dt; h->sim_tend = h->traj_t_end;
    }
    for (int k = 0; k < steps; k++) {
        if (state_hist_dev) HIPCHK(h, hipMemcpyAsync(state_hist_dev + (size_t)k * B * 6, h->d_state, (size_t)B * 6 * sizeof(real), hipMemcpyDeviceToDevice, h->stream));      // push!(qs, state) :88
        if (control_hist_dev) HIPCHK(h, hipMemcpyAsync(control_hist_dev + (size_t)k * B * 3, h->d_control, (size_t)B * 3 * sizeof(real), hipMemcpyDeviceToDevice, h->stream)); // push!(us, control) :89
        if ((rc = launch_nodes(h, true)) || (rc = update_and_solve(h, nullptr))) return rc;          // :90-93 (time grid fused into the projection launch)
        h->sim_idx++;                                                                                                                                         // (t0 now holds element sim_idx of the clock)
        hipLaunchKernelGGL(k_advance, dim3((B + 63) / 64), dim3(64), 0, h->stream, h->dc, B, dt, h->d_state, h->d_control, h->d_u, h->d_t0, h->d_tstart, h->sim_clk, h->sim_idx);      // :94-95
        LAUNCH_CHECK(h);
    }
    return PG_OK;
}
int pg_simulate_clock(pg_handle* h, double dt, int32_t steps, int32_t B, const double* t_start, double* out) {
#pragma clang fp contract(off)
    if (!h) return PG_ERR_INVALID;
    REQUIRE(h, steps >= 1 && dt > 0.0 && B >= 1 && t_start && out, "pg_simulate_clock: steps >= 1, dt > 0, B >= 1 and both arrays required");
    REQUIRE(h, h->d_traj, "pg_simulate_clock: no trajectory installed (its last time is the stop of the range)");
    const JlRange clk = jl_colon(0.0, dt, h->traj_t_end);
    for (int b = 0; b < B; b++) {
        double acc = t_start[b];
        for (int k = 0; k < steps; k++) { out[(size_t)k * B + b] = h->dc.time_grid_naive ? acc : jl_shifted_elem(clk, t_start[b], k + 1); acc = acc + dt; }
    }
    return PG_OK;
}
int pg_get_state(pg_handle* h, double* state, double* control, double* t0) {
    int rc = check_ready(h); if (rc) return rc;
    const size_t B = h->B;
    if ((rc = down(h, state, h->d_state, B * 6)) || (rc = down(h, control, h->d_control, B * 3)) || (rc = down_raw(h, t0, h->d_t0, B * 8))) return rc;
    return PG_OK;
}
int pg_get_phase_ms(pg_handle* h, float out3[3]) {
    if (!h || !out3) return PG_ERR_INVALID;
    if (!h->timing_valid) {
        h->err = h->phase_timing == 0 ? "phase timing is off (the default since round 5): pg_set_option(h, \"phase_timing\", 1), then pg_step_dev" : "no pg_step_dev recorded yet";
        return PG_ERR_STATE;
    }
    HIPCHK(h, hipEventSynchronize(h->ev[3]));
    for (int i = 0; i < 3; i++) HIPCHK(h, hipEventElapsedTime(&out3[i], h->ev[i], h->ev[i + 1]));
    return PG_OK;
}
// the five input arrays in the staging buffer, in the layout of the device block (batch == capacity)
static void stage_block(pg_handle* h, int32_t B, const double* s_, const double* c_, const double* t0, const double* o_, const double* toff) {
    real* st = (real*)h->h_stage; real* ct = st + (size_t)B * 6; real* ot = ct + (size_t)B * 3; double* tt = (double*)(h->h_stage + h->in_dbl_off); double* ft = tt + B;
    for (size_t i = 0; i < (size_t)B * 6; i++) st[i] = (real)s_[i];
    for (size_t i = 0; i < (size_t)B * 3; i++) ct[i] = (real)c_[i];
    for (size_t i = 0; i < (size_t)B * 4; i++) ot[i] = o_ ? (real)o_[i] : real(0.0);
    memcpy(tt, t0, (size_t)B * 8);
    if (toff) memcpy(ft, toff, (size_t)B * 8); else memset(ft, 0xFF, (size_t)B * 8);      // all-ones bit pattern is a NaN: path-tracking mode
}
// pg_step of a small warm batch as ONE graph launch.  Returns PG_OK with *done = true when the step ran through the graph; *done = false means "not eligible / not
// possible": the caller takes the ordinary path.  Eligible: graphs on, the batch fills the handle (one copy per direction), <= 256 instances (beyond, the kernels
// dominate the launches), every instance warm (the cold branch picks other kernels), nothing fused or chunked.
static int step_by_graph(pg_handle* h, int32_t B, const double* state, const double* control, const double* t0, const double* other, const double* toff, bool* done) {
    *done = false;
    auto& G = h->sg;
    if (!h->graph_mode || G.disabled || B != h->cfg.batch_capacity || B > 256 || h->B != B || h->warm_B < B || h->fuse != 0 || !h->d_traj) return PG_OK;
    if (h->dc.n_traj > 1 && h->traj_idx_B < B) return PG_OK;
    if (hipSetDevice(h->cfg.device) != hipSuccess) return PG_OK;
    const bool same = G.x && G.B == B && G.user == h->stream && G.fuse == h->fuse && G.pipeline == h->pipeline && G.has_hji == (int)h->has_hji && G.traj_L == h->traj_L &&
                      memcmp(&G.dc, &h->dc, sizeof(DevCfg)) == 0 && memcmp(&G.hv, &h->hv, sizeof(HjiView)) == 0;
    hipStream_t run = h->stream ? h->stream : G.own;
    if (!same) {
        if (G.x) { (void)hipGraphExecDestroy(G.x); G.x = nullptr; }
        if (G.g) { (void)hipGraphDestroy(G.g); G.g = nullptr; }
        if (!h->stream && !G.own && hipStreamCreate(&G.own) != hipSuccess) { G.disabled = true; (void)hipGetLastError(); return PG_OK; }      // (a blocking stream: ordered against the null stream)
        run = h->stream ? h->stream : G.own;
        if (hipStreamSynchronize(h->stream) != hipSuccess) return PG_OK;
        hipStream_t user = h->stream;
        bool ok = hipStreamBeginCapture(run, hipStreamCaptureModeThreadLocal) == hipSuccess;
        if (ok) {
            h->stream = run; G.capturing = true;
            ok = hipMemcpyAsync(h->d_in, h->h_stage, h->in_bytes, hipMemcpyHostToDevice, run) == hipSuccess && pg_step_dev(h, nullptr) == PG_OK &&
                 hipMemcpyAsync(h->h_stage, h->d_out, h->out_bytes, hipMemcpyDeviceToHost, run) == hipSuccess;
            h->stream = user; G.capturing = false;
            hipGraph_t g = nullptr;
            const bool ended = hipStreamEndCapture(run, &g) == hipSuccess && g != nullptr;
            ok = ok && ended;
            if (ok) { G.g = g; ok = hipGraphInstantiate(&G.x, G.g, nullptr, nullptr, 0) == hipSuccess; }
            else if (g) (void)hipGraphDestroy(g);
        }
        if (!ok) {           // this runtime / this configuration cannot be captured: never again for this handle, the ordinary path takes over
            if (G.x) { (void)hipGraphExecDestroy(G.x); G.x = nullptr; }
            if (G.g) { (void)hipGraphDestroy(G.g); G.g = nullptr; }
            G.disabled = true; (void)hipGetLastError();
            return PG_OK;
        }
        memcpy(&G.dc, &h->dc, sizeof(DevCfg)); memcpy(&G.hv, &h->hv, sizeof(HjiView)); G.B = B;      // (byte copies: the signature is compared with memcmp, padding included)
         G.user = user; G.fuse = h->fuse; G.pipeline = h->pipeline; G.has_hji = (int)h->has_hji; G.traj_L = h->traj_L;
    } else if (hipStreamSynchronize(h->stream) != hipSuccess) return PG_OK;
    stage_block(h, B, state, control, t0, other, toff);
    HIPCHK(h, hipGraphLaunch(G.x, run));
    HIPCHK(h, hipStreamSynchronize(run));
    h->timing_valid = false;
    *done = true;
    return PG_OK;
}
int pg_step(pg_handle* h, int32_t B, const double* state, const double* control, const double* t0, const double* other, const double* toff,
            double* u_out, int32_t* status, int32_t* iters) {
    if (h && state && control && t0) {
        bool done = false;
        int rc = step_by_graph(h, B, state, control, t0, other, toff, &done); if (rc) return rc;
        if (done) {
            const real* us = (const real*)h->h_stage; const int* ss = (const int*)(us + (size_t)B * 3); const int* is = ss + B;
            if (u_out) for (size_t i = 0; i < (size_t)B * 3; i++) u_out[i] = (double)us[i];
            if (status) memcpy(status, ss, (size_t)B * sizeof(int));
            if (iters) memcpy(iters, is, (size_t)B * sizeof(int));
            return PG_OK;
        }
    }
    int rc = pg_set_inputs(h, B, state, control, t0, other, toff); if (rc) return rc;
    if ((rc = pg_step_dev(h, nullptr))) return rc;
    // controls, status and iteration counts come back through the pinned staging buffer: three asynchronous copies behind k_solve on the handle's stream
    // (ordered even when the caller installed a non-blocking stream), ONE synchronisation, then the conversion into the caller's arrays
    real* us = (real*)h->h_stage; int* ss = (int*)(us + (size_t)B * 3); int* is = ss + B;
    if (B == h->cfg.batch_capacity) {          // (u, status, iters) are one block on the device: one copy
        HIPCHK(h, hipMemcpyAsync(h->h_stage, h->d_out, h->out_bytes, hipMemcpyDeviceToHost, h->stream));
    } else {
        if (u_out) HIPCHK(h, hipMemcpyAsync(us, h->d_u, (size_t)B * 3 * sizeof(real), hipMemcpyDeviceToHost, h->stream));
        if (status) HIPCHK(h, hipMemcpyAsync(ss, h->d_status, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, h->stream));
        if (iters) HIPCHK(h, hipMemcpyAsync(is, h->d_iters, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (u_out) for (size_t i = 0; i < (size_t)B * 3; i++) u_out[i] = (double)us[i];
    if (status) memcpy(status, ss, (size_t)B * sizeof(int));
    if (iters) memcpy(iters, is, (size_t)B * sizeof(int));
    return PG_OK;
}

// ---- read-backs --------------------------------------------------------------------------------------------------
int pg_get_time_steps(pg_handle* h, double* ts, double* dt, double* prev_ts) {
    int rc = check_ready(h); if (rc) return rc;
    const size_t B = h->B; const DevCfg& C = h->dc;
    if ((rc = down_raw(h, ts, h->d_ts, B * C.NN * 8)) || (rc = down_raw(h, dt, h->d_dt, B * C.N * 8)) || (rc = down_raw(h, prev_ts, h->d_prev_ts, B * C.NN * 8))) return rc;
    return PG_OK;
}
int pg_get_nodes(pg_handle* h, double* qs, double* us, double* ps) {
    int rc = check_ready(h); if (rc) return rc;
    const size_t B = h->B; const DevCfg& C = h->dc;
    std::vector<double> nd(B * C.NN * 10);
    if ((rc = down(h, nd.data(), h->d_nodes, nd.size()))) return rc;
    for (size_t i = 0; i < B * C.NN; i++) {
        const double* r = &nd[i * 10];
        if (qs) for (int k = 0; k < 6; k++) qs[i * 6 + k] = r[k];
        if (us) { us[i * 2] = r[6]; us[i * 2 + 1] = r[7]; }
        if (ps) { ps[i * 4] = r[8]; ps[i * 4 + 1] = r[9]; ps[i * 4 + 2] = 0.0; ps[i * 4 + 3] = 0.0; }
    }
    return PG_OK;
}
int pg_get_path_coordinates(pg_handle* h, double* sep) {
    int rc = check_ready(h); if (rc) return rc;
    REQUIRE(h, sep, "sep is null");
    const size_t B = h->B;
    std::vector<double> s4(B * 4);
    if ((rc = down(h, s4.data(), h->d_sep, s4.size()))) return rc;
    for (size_t b = 0; b < B; b++) for (int k = 0; k < 3; k++) sep[b * 3 + k] = s4[b * 4 + k];
    return PG_OK;
}
int pg_get_qp(pg_handle* h, int32_t b0, int32_t n, double* out) {
    int rc = check_ready(h); if (rc) return rc;
    REQUIRE(h, out && b0 >= 0 && n >= 1 && b0 + n <= h->B, "pg_get_qp: range outside the batch");
    if (h->dc.formulation == PG_DECOUPLED && !h->qp_embedded) {      // the last update_QP! of this lateral handle wrote the packed records only: the embedded block now, from the same nodes
        const long nt = (long)h->B * h->dc.N;
        hipLaunchKernelGGL(k_qp_dec, dim3((unsigned)((nt + 127) / 128)), dim3(128), 0, h->stream, h->dc, h->B, h->d_nodes, h->d_dt, h->d_qp, 1);
        LAUNCH_CHECK(h);
        h->qp_embedded = true;
    }
    return down(h, out, h->d_qp + (size_t)b0 * h->dc.qp_len, (size_t)n * h->dc.qp_len);
}
int pg_set_qp(pg_handle* h, int32_t b0, int32_t n, const double* in) {
    int rc = check_ready(h); if (rc) return rc;
    REQUIRE(h, in && b0 >= 0 && n >= 1 && b0 + n <= h->B, "pg_set_qp: range outside the batch");
    if (h->dc.formulation == PG_DECOUPLED && !h->qp_embedded) {      // (a partial install on top of a step that wrote the packed records only: complete the embedded block first)
        const long nt_ = (long)h->B * h->dc.N;
        hipLaunchKernelGGL(k_qp_dec, dim3((unsigned)((nt_ + 127) / 128)), dim3(128), 0, h->stream, h->dc, h->B, h->d_nodes, h->d_dt, h->d_qp, 1);
        LAUNCH_CHECK(h);
        h->qp_embedded = true;
    }
    if ((rc = up(h, h->d_qp + (size_t)b0 * h->dc.qp_len, in, (size_t)n * h->dc.qp_len))) return rc;
    if (h->solve_lat) {      // k_solve_lat reads the packed stage records: refresh them from the installed block
        const long nt = (long)n * h->dc.N;
        hipLaunchKernelGGL(k_lat_pack, dim3((unsigned)((nt + 127) / 128)), dim3(128), 0, h->stream, h->dc, b0, n, h->d_qp);
        LAUNCH_CHECK(h);
    }
    if (b0 == 0 && n == h->B) h->qp_stale = false;      // (both record forms of the whole batch are current now)
    return PG_OK;
}
int pg_get_solution(pg_handle* h, double* x, double* sigma) {
    int rc = check_ready(h); if (rc) return rc;
    const size_t B = h->B; const DevCfg& C = h->dc;
    if ((rc = down(h, x, h->d_solx, B * C.NN * 8)) || (rc = down(h, sigma, h->d_sigma, B * C.N * 3))) return rc;
    return PG_OK;
}
int pg_get_solve_info(pg_handle* h, int32_t* status, int32_t* iters, uint16_t* active, double* mu) {
    int rc = check_ready(h); if (rc) return rc;
    const size_t B = h->B; const DevCfg& C = h->dc;
    if ((rc = down_raw(h, status, h->d_status, B * 4)) || (rc = down_raw(h, iters, h->d_iters, B * 4)) || (rc = down_raw(h, active, h->d_active, B * C.N * 2)) ||
        (rc = down(h, mu, h->d_mu, B))) return rc;
    return PG_OK;
}
int pg_get_polish_info(pg_handle* h, int32_t* polish) {
    int rc = check_ready(h); if (rc) return rc;
    REQUIRE(h, polish, "polish is null");
    return down_raw(h, polish, h->d_polish, (size_t)h->B * 4);
}
int pg_get_multipliers(pg_handle* h, double* lam) {
    int rc = check_ready(h); if (rc) return rc;
    REQUIRE(h, lam, "lam is null");
    return down(h, lam, h->d_lam, (size_t)h->B * h->dc.N * 16);
}
int pg_get_walls(pg_handle* h, double* edges) {
    int rc = check_ready(h); if (rc) return rc;
    if (!h->dc.walls) { h->err = "walls are off (pg_config.walls = 0)"; return PG_ERR_STATE; }
    REQUIRE(h, edges, "edges is null");
    return down(h, edges, h->d_walls, (size_t)h->B * h->dc.N * 2);
}
int pg_get_hji_constraint(pg_handle* h, double* M, double* b, double* V) {
    int rc = check_ready(h); if (rc) return rc;
    const size_t B = h->B;
    std::vector<double> mb(B * 4);
    if (h->has_hji && (rc = down(h, mb.data(), h->d_Mb, mb.size()))) return rc;
    for (size_t i = 0; i < B; i++) {
        if (!h->has_hji) { mb[4 * i] = 0; mb[4 * i + 1] = 0; mb[4 * i + 2] = 1; mb[4 * i + 3] = INFINITY; }
        if (M) { M[2 * i] = mb[4 * i]; M[2 * i + 1] = mb[4 * i + 1]; }
        if (b) b[i] = mb[4 * i + 2];
        if (V) V[i] = mb[4 * i + 3];
    }
    return PG_OK;
}

int pg_hji_lookup_dev(pg_handle* h, int32_t B, const void* x7_dev, void* V_dev, void* gradV_dev) {
    if (!h) return PG_ERR_INVALID;
    REQUIRE(h, h->has_hji, "no HJI grid installed");
    REQUIRE(h, B >= 1 && x7_dev, "pg_hji_lookup_dev: bad arguments");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    real* out8 = nullptr; const size_t R = sizeof(real);
    HIPCHK(h, hipMalloc((void**)&out8, (size_t)B * 8 * R));
    int rc = launch_hji_lookup(h, B, (const real*)x7_dev, out8);
    if (!rc) {
        if (V_dev) HIPCHK(h, hipMemcpy2DAsync(V_dev, R, out8, 8 * R, R, B, hipMemcpyDeviceToDevice, h->stream));
        if (gradV_dev) HIPCHK(h, hipMemcpy2DAsync(gradV_dev, 7 * R, out8 + 1, 8 * R, 7 * R, B, hipMemcpyDeviceToDevice, h->stream));
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    (void)hipFree(out8);
    return rc;
}
int pg_hji_lookup8_dev(pg_handle* h, int32_t B, const void* x7_dev, void* out8_dev) {
    if (!h) return PG_ERR_INVALID;
    REQUIRE(h, h->has_hji, "no HJI grid installed");
    REQUIRE(h, B >= 1 && x7_dev && out8_dev, "pg_hji_lookup8_dev: bad arguments");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    return launch_hji_lookup(h, B, (const real*)x7_dev, (real*)out8_dev);
}
int pg_hji_grid_dims(pg_handle* h, int32_t dims[7]) {
    if (!h || !dims) return PG_ERR_INVALID;
    REQUIRE(h, h->has_hji, "no HJI grid installed");
    for (int d = 0; d < 7; d++) dims[d] = h->hv.dims[d];
    return PG_OK;
}
int pg_hji_slice(pg_handle* h, int32_t B, const double* q7, double* V_out, double* rgb_out, double* cross_x, double* cross_y) {
    if (!h) return PG_ERR_INVALID;
    REQUIRE(h, h->has_hji, "no HJI grid installed");
    REQUIRE(h, B >= 1 && q7 && V_out, "pg_hji_slice: need B >= 1, the relative states and V_out");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const int n1 = h->hv.dims[0], n2 = h->hv.dims[1]; const size_t npt = (size_t)B * n1 * n2;
    real *dq = nullptr, *dx = nullptr, *dv = nullptr, *dV = nullptr, *drgb = nullptr, *dcx = nullptr, *dcy = nullptr;
    int rc = PG_OK;
    auto cleanup = [&]() { for (void* p : {(void*)dq, (void*)dx, (void*)dv, (void*)dV, (void*)drgb, (void*)dcx, (void*)dcy}) if (p) (void)hipFree(p); };
#define SL_ALLOC(ptr, count) do { if (hipMalloc((void**)&(ptr), (size_t)(count) * sizeof(real)) != hipSuccess) { h->err = "pg_hji_slice: hipMalloc failed"; cleanup(); return PG_ERR_HIP; } } while (0)
    SL_ALLOC(dq, (size_t)B * 7); SL_ALLOC(dx, npt * 7); SL_ALLOC(dv, npt * 8); SL_ALLOC(dV, npt);
    if (rgb_out) SL_ALLOC(drgb, npt * 3);
    if (cross_x) SL_ALLOC(dcx, (size_t)B * (n1 - 1) * n2);
    if (cross_y) SL_ALLOC(dcy, (size_t)B * n1 * (n2 - 1));
#undef SL_ALLOC
    if ((rc = up(h, dq, q7, (size_t)B * 7))) { cleanup(); return rc; }
    const dim3 grid((unsigned)((npt + 255) / 256));
    hipLaunchKernelGGL(k_hji_slice_queries, grid, dim3(256), 0, h->stream, h->hv, B, dq, dx);
    if ((rc = launch_hji_lookup(h, (int)npt, dx, dv))) { cleanup(); return rc; }
    hipLaunchKernelGGL(k_hji_slice_post, grid, dim3(256), 0, h->stream, h->hv, B, dv, dV, drgb, dcx, dcy);
    if (hipGetLastError() != hipSuccess) { h->err = "pg_hji_slice: kernel launch failed"; cleanup(); return PG_ERR_HIP; }
    if ((rc = down(h, V_out, dV, npt)) || (rc = down(h, rgb_out, drgb, npt * 3)) || (rc = down(h, cross_x, dcx, (size_t)B * (n1 - 1) * n2)) ||
        (rc = down(h, cross_y, dcy, (size_t)B * n1 * (n2 - 1)))) { cleanup(); return rc; }
    cleanup();
    return PG_OK;
}
int pg_hji_lookup(pg_handle* h, int32_t B, const double* x7, double* V, double* gradV) {
    if (!h) return PG_ERR_INVALID;
    REQUIRE(h, h->has_hji, "no HJI grid installed");
    REQUIRE(h, B >= 1 && x7, "pg_hji_lookup: bad arguments");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    real *dx = nullptr, *dout = nullptr;
    HIPCHK(h, hipMalloc((void**)&dx, (size_t)B * 7 * sizeof(real)));
    HIPCHK(h, hipMalloc((void**)&dout, (size_t)B * 8 * sizeof(real)));
    int rc = up(h, dx, x7, (size_t)B * 7);
    if (!rc) rc = launch_hji_lookup(h, B, dx, dout);
    std::vector<double> out((size_t)B * 8);
    if (!rc) rc = down(h, out.data(), dout, out.size());
    if (!rc) {
        for (int i = 0; i < B; i++) { if (V) V[i] = out[8 * (size_t)i]; if (gradV) for (int k = 0; k < 7; k++) gradV[7 * (size_t)i + k] = out[8 * (size_t)i + 1 + k]; }
    }
    (void)hipFree(dx); (void)hipFree(dout);
    return rc;
}

}  // extern "C"
