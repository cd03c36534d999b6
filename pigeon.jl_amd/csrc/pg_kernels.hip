// HIP kernels of the batched MPC hot path for gfx950 (MI355X), fp64.
//   k_time_steps    T1   model_predictive_control.jl:17-30          lane  = instance
//   k_project       P1   trajectories.jl:71-94, math.jl:4-9         wave  = instance (64 lanes x segments, wave arg-min)
//   k_nodes         N1/N2 coupled_lat_long.jl:62-142                lane  = instance (serial (V,s) recurrence)
//   k_linearize     L1/L2 coupled_lat_long.jl:335-353 (+linearize)  lane  = (instance, interval, tangent pair)
//   k_limits        Q2   coupled_lat_long.jl:323-333,354-367        lane  = (instance, interval)
//   k_hji_*         H1-H5 HJI_computation.jl:20-24,66-131,160-170   wave  = lookup (coalesced 64 B gathers)
//   k_solve         Q3/Q4 solve! + get_next_control                 wave  = QP instance (Riccati interior point, LDS staged)
// Data layout notes are in DESIGN.md.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "pg_device.hpp"
#include "pg_julia_range.hpp"

namespace pg {

struct DevCfg {
    DevVehicle veh;
    DevControl cp;
    int Ns, Nl, N, NN;            // NN = N + 1 nodes
    tdouble dt_short, dt_long;
    int use_correction_step, nsub;
    int formulation;              // PG_COUPLED / PG_DECOUPLED
    int dbg_poison;               // diagnostic kernel build only (option "diag_lat_poison"): k_solve_lat fills its LDS with NaN before it starts and the host its workspaces before every launch -- a read before the launch's own write shows
    int dbg_instance;             // diagnostic kernel build only: instance whose interior-point trace is printed (-1 = none; option "diag_instance" of the -DPG_DIAG build)
    real ux_dummy;              // decoupled: value of the inert Ux slot of the embedded 8-state problem (strictly inside [V_min, V_max])
    int alias_prev_ts;            // the reference's MPCTimeSteps passes `ts` as prev_ts too (model_predictive_control.jl:15): same array
    JlRange tg_short, tg_long;    // dt_short*(0:Ns), dt_long*(1:Nl) as Julia builds them (model_predictive_control.jl:25-26; pg_julia_range.hpp), made by pg_create
    int time_grid_naive;          // option "time_grid_naive": 1 = t0 + dt*i with two roundings (rounds 1-5) instead of the range arithmetic; also t += dt in the closed loop
    int has_hji;
    real hji_eps;
    real un0, un1;              // u_normalization (coupled_lat_long.jl:199)
    real fxmin_n;               // Fx_min / un1
    int qp_len;
    int ipm_max_iter;
    real ipm_tol, ipm_mu0;
    int polish;                   // active-set polish after the interior point (k_solve): 0 off, 1 on
    real polish_rho, polish_tol;  // penalty of the active rows in the polish solves; feasibility tolerance of its verification
    int hji_rounds;               // k_solve: working sets such a seeded attempt may try (0: the cold-guess cap and its extension)
    int warm_trivial_cold;        // k_solve: a warm instance whose previous working set was empty starts like a cold one (option "warm_trivial_cold", default 1)
    int ck_riccati;               // k_solve (rounds-only instantiation): restart the matrix recursion of a round at its checkpoint when the working set allows (option "ck_riccati", default 1)
    int clip_stops;               // k_solve: ... and the steering angle at its stops (option "clip_stops")
    int clip_guess;               // k_solve, cold instances: the first roll-out clips the steering rate at its limits and the clipped transitions are the first working set (option "clip_guess", default 1)
    int hji_seed;                 // k_solve: rounds of an instance whose safety row is violated at the current control start from a seeded working set (0: interior point, as before)
    int cold_guess;               // > 0: a COLD instance first tries the polish from the empty active set (unconstrained LQ optimum + add/drop rounds), at most this many rounds
    int warm_polish;              // instances with a previous solution first try the polish from its active set and multipliers (no interior point if it verifies)
    real polish_ipm_tol;          // interior-point tolerance at which the polish is first attempted (>= ipm_tol; a polish that fails there resumes the interior point down to ipm_tol)
    TrajView traj;                // trajectory 0 of the installed library
    int n_traj;                   // library size (1: every instance tracks `traj`)
    long traj_stride;             // doubles between consecutive trajectories of the library ([n_traj][10][Lmax])
    const int* traj_idx;          // [B] per-instance selection (nullptr when n_traj == 1)
    const int* traj_len;          // [n_traj] valid nodes of each trajectory
    int walls;                    // build-defined extension: soft rows edge_R - sw <= e <= edge_L + sw, sw >= 0 at nodes 2..N+1 (decoupled formulation only)
    real wall_weight;           // linear penalty on sw (per second, like W_beta)
    real* wall_edges;           // [B][N][2] (edge_L, edge_R) at node k+1, written by k_nodes_dec, read by k_solve
    int lat_pin;                // k_solve_lat: a held steering-rate row pins the input of its stage exactly (1; 0 = every held row through the augmented Lagrangian, as in round 4)
    int lat_polish2;            // k_solve_lat: the polish gets a second chance behind the resumed interior point
    real lat_far_cost;          // k_solve_lat: starting cost per row beyond which the early hand-over to the polish is not tried
    real lat_mu0_cost;          // k_solve_lat: first barrier parameter = max(ipm_mu0, lat_mu0_cost x cost of the starting point per row)
    real lat_rho_scale;         // k_solve_lat: penalty of the held rows = polish_rho x this (see pg_solve_lat.hip)
    int lat_polish_rounds;      // k_solve_lat: working sets a polish may try at the hand-over tolerance (one fewer behind the resumed interior point)
    int lat_settle;             // k_solve_lat: working-set decisions wait for settled multipliers (0 never, 1 warm attempts, 2 every polish)
    int lat_wipm; real lat_wmu, lat_wtau;      // k_solve_lat: interior point of a warm instance starts from the previous solution (floors of t lambda and of t)
    int lat_warm_rounds;        // k_solve_lat: working sets a warm attempt (previous step's set and multipliers) may try before the cold start takes over
    int nodes_serial;           // option "nodes_serial" (A/B and parity tests): 1 = the cold seeding commits ONE node per pass -- the serial recurrence of the reference to the last bit
    int lat_aux_gate;           // k_solve_lat (option "lat_aux_gate"): the serial passes leave F / B'PB / B'y in lat_aux only while an instance of the wavefront is in a polish
    real* lat_spc;              // lateral formulation: k_solve_lat's lane-contiguous copy of the stage constants of its stage-parallel passes (LAT_SPC_Q; nullptr: not wanted)
    const real* lat_zero;       // eight stored zeros behind lat_aux: the row a lane of k_solve_lat's roll-out reads when its side of the row is empty
    real* lat_aux;              // lateral formulation: [B][64][8] what the multiplier of a pinned rate row is read from (k_solve_lat, see its header)
    char* lat_ws;               // lateral formulation, horizons beyond 32 intervals: k_solve_lat's per-wavefront workspace (lat_ws_bytes(B); nullptr: not wanted)
    real* lat_pack;             // lateral formulation: [B][N][LATP] packed stage records for k_solve_lat, written by k_qp_dec next to the QP block (nullptr: not wanted)
};
// the reference trajectory instance b tracks (mpc.trajectory of that controller)
PG_DEV TrajView traj_of(const DevCfg& C, int b) {
    TrajView T = C.traj;
    if (C.n_traj > 1) {
        const int k = C.traj_idx[b]; const long off = (long)k * C.traj_stride;
        T.L = C.traj_len[k];
        T.t += off; T.s += off; T.V += off; T.A += off; T.E += off; T.N += off; T.psi += off; T.kappa += off; T.edge_L += off; T.edge_R += off;
    }
    return T;
}

// Stage block as k_solve keeps it in LDS (packed from the QP data on the way in): rows 0..5 of Abar = [A | B0+Bf] at a row stride of 9 doubles (odd stride: the 8 rows land on distinct LDS
// banks, so the row-indexed reads of the Riccati passes are conflict-free), then Bbar = Bf (12) at SB_B, cbar = c (6) at SB_C.
constexpr int SB = 72, SB_ROW = 9, SB_B = 54, SB_C = 66;

// Packed stage record of the LATERAL formulation (k_qp_dec writes it, k_solve_lat streams it from L2: 16-byte aligned rows, one 64 B line per matrix row):
//   [8 i + m], i < 4, m < 7: row i of [A | B0+Bf | Bf | c]  (= rows 0..3 of [Abar | Bbar | cbar]; m = 7 is a stored 0: the column lanes 7..15 of a row read)
//   [32..39] H (4 x 2, row-major)   [40..43] G   [44] dmax  [45] dmin  [46] ddmax  [47] ddmin  [48] dt  [49..55] 0
constexpr int LATP = 56;

// offsets inside one instance's QP block (doubles); same order as pg_get_qp documents
struct QpOff { int A, B0, Bf, c, H, G, dmin, dmax, fxmax, ddmin, ddmax, dt, qcurr, ucurr, M, b; };
__host__ __device__ inline QpOff qp_offsets(int N) {
    QpOff o; int p = 0;
    o.A = p; p += 36 * N; o.B0 = p; p += 12 * N; o.Bf = p; p += 12 * N; o.c = p; p += 6 * N; o.H = p; p += 8 * N; o.G = p; p += 4 * N;
    o.dmin = p; p += N; o.dmax = p; p += N; o.fxmax = p; p += N; o.ddmin = p; p += N; o.ddmax = p; p += N; o.dt = p; p += N;
    o.qcurr = p; p += 6; o.ucurr = p; p += 2; o.M = p; p += 2; o.b = p; p += 1;
    return o;
}

// ------------------------------------------------------------------------------------------------------------------
// absolute time stays fp64 in both builds (tdouble)
__global__ __launch_bounds__(64) void k_time_steps(DevCfg C, int B, const double* __restrict__ t0, double* __restrict__ ts, double* __restrict__ dt, double* __restrict__ prev_ts) {
#pragma clang fp contract(off)   // the time grid is compared bit-for-bit with the CPU restatement: no fused multiply-add here
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double* T = ts + (size_t)b * C.NN; double* PT = prev_ts + (size_t)b * C.NN; double* D = dt + (size_t)b * C.N;
    for (int i = 0; i < C.NN; i++) PT[i] = T[i];                                   // :20
    double t = t0[b];
    double t0_long = t + C.Ns * C.dt_short;                                        // :21
    if (C.use_correction_step) t0_long = C.dt_long * ceil((t0_long + C.dt_short) / C.dt_long - 1.0);   // :23
    for (int i = 0; i <= C.Ns; i++) T[i] = C.time_grid_naive ? t + C.dt_short * i : jl_shifted_elem(C.tg_short, t, i + 1);                     // :25  t0 .+ dt_short*(0:N_short)
    for (int i = 1; i <= C.Nl; i++) T[C.Ns + i] = C.time_grid_naive ? t0_long + C.dt_long * i : jl_shifted_elem(C.tg_long, t0_long, i);       // :26  t0_long .+ dt_long*(1:N_long)
    for (int i = 0; i < C.N; i++) D[i] = T[i + 1] - T[i];                          // :27-29
    if (C.alias_prev_ts) for (int i = 0; i < C.NN; i++) PT[i] = T[i];              // prev_ts IS ts in the reference (:15)
}

// ------------------------------------------------------------------------------------------------------------------
// math.jl:4-9
PG_DEV real seg_dist2(real ax, real ay, real bx, real by, real x, real y) {
    real vx = bx - ax, vy = by - ay;
    real lam = (vx * (x - ax) + vy * (y - ay)) / (vx * vx + vy * vy);
    lam = lam < real(0.0) ? real(0.0) : (lam > real(1.0) ? real(1.0) : lam);
    real px = (real(1.0) - lam) * ax + lam * bx, py = (real(1.0) - lam) * ay + lam * by;
    return (px - x) * (px - x) + (py - y) * (py - y);
}
// compute_time_steps! for lane i = node i of one instance (the same statements as k_time_steps, one node per lane): used by the fused launch of pg_step_dev
PG_DEV void time_grid_lane(const DevCfg& C, int i, double t, double* T, double* D, double* PT) {
#pragma clang fp contract(off)   // bit-for-bit with the CPU restatement: no fused multiply-add
    if (i >= C.NN) return;
    PT[i] = T[i];                                                                   // :20
    double t0_long = t + C.Ns * C.dt_short;                                         // :21
    if (C.use_correction_step) t0_long = C.dt_long * ceil((t0_long + C.dt_short) / C.dt_long - 1.0);   // :23
    auto node_time = [&](int k) __attribute__((always_inline)) -> double {                         // :25-26 (Julia's range elements: pg_julia_range.hpp)
        if (C.time_grid_naive) return k <= C.Ns ? t + C.dt_short * k : t0_long + C.dt_long * (k - C.Ns);
        return k <= C.Ns ? jl_shifted_elem(C.tg_short, t, k + 1) : jl_shifted_elem(C.tg_long, t0_long, k - C.Ns);
    };
    const double Ti = node_time(i);
    const int n = i + 1;
    const double Tn = node_time(n);
    T[i] = Ti;
    if (i < C.N) D[i] = Tn - Ti;                                                    // :27-29
    if (C.alias_prev_ts) PT[i] = Ti;                                                // prev_ts IS ts in the reference (:15)
}
// one wave per instance; strict '<' with lowest index winning ties == the reference's sequential scan (:71-79).  With t0 != nullptr the same wave also
// writes the instance's time grid (lane = node): pg_step_dev launches time grid + projection as one kernel (TG = true; the fp32 purity check allows fp64
// arithmetic in that instantiation only).
template <bool TG> __global__ __launch_bounds__(256) void k_project(DevCfg C, int B, const real* __restrict__ state, real* __restrict__ sep, const double* __restrict__ t0, double* __restrict__ ts,
                                                 double* __restrict__ dt, double* __restrict__ prev_ts, int* __restrict__ progress = nullptr, int n_progress = 0, int* __restrict__ order_cnt = nullptr,
                                                 int* __restrict__ solve_ctl = nullptr, int ctl_parity = 0) {
    int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (progress && blockIdx.x == 0) { for (int i = threadIdx.x; i < n_progress; i += blockDim.x) progress[i] = 0; }      // (k_nodes_linearize of this step counts from 0)
    if (order_cnt && blockIdx.x == 0 && threadIdx.x < 2) order_cnt[threadIdx.x] = 0;                                       // (the launch order the nodes kernel files: both counters from 0)
    if (solve_ctl && blockIdx.x == 0 && threadIdx.x == 2) solve_ctl[ctl_parity] = 0;                                        // (the to-do counter of this step's solve launch)
    if (wave >= B) return;
    if constexpr (TG) time_grid_lane(C, lane, t0[wave], ts + (size_t)wave * C.NN, dt + (size_t)wave * C.N, prev_ts + (size_t)wave * C.NN);
    const TrajView T = traj_of(C, wave);
    real x = state[(size_t)wave * 6 + 0], y = state[(size_t)wave * 6 + 1];
    real best = INFINITY; int bi = 0x7fffffff;
    for (int i = lane; i < T.L - 1; i += 64) {
        real d2 = seg_dist2(T.E[i], T.N[i], T.E[i + 1], T.N[i + 1], x, y);
        if (d2 < best) { best = d2; bi = i; }           // i increases within a lane, so strict '<' keeps the lowest index
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        real ob = __shfl_xor(best, off); int oi = __shfl_xor(bi, off);
        if (ob < best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) {
        int i = bi;
        real* o = sep + (size_t)wave * 4;
        // non-finite pose: every distance is NaN/Inf, no segment ever wins and bi keeps its sentinel.  The reference indexes traj[0] there (BoundsError,
        // caught by the ROS loop, ros_integration.jl:95-102); here the instance is poisoned with NaN so that k_solve reports PG_NUMERICAL and no lane
        // ever forms an address from the sentinel
        if (i < 0 || i > T.L - 2) { o[0] = NAN; o[1] = NAN; o[2] = NAN; o[3] = real(0.0); return; }
        real vx = T.E[i + 1] - T.E[i], vy = T.N[i + 1] - T.N[i], wx = x - T.E[i], wy = y - T.N[i];
        // :82 sqrt(w.w - d^2) with d^2 = |w - lam v|^2 expanded: w.w - d^2 = lam (2 w.v - lam v.v).  Same value for every lam in [0,1], but no
        // cancellation (the literal form loses half the digits when the foot point is near the segment start: 5 mm in fp32)
        real vw = vx * wx + vy * wy, vv = vx * vx + vy * vy;
        real lam = vw / vv; lam = lam < real(0.0) ? real(0.0) : (lam > real(1.0) ? real(1.0) : lam);
        real ds = sqrt(fmax(lam * (real(2.0) * vw - lam * vv), real(0.0)));
        real cr = vx * wy - vy * wx;
        real Ai = (T.V[i + 1] - T.V[i]) / (T.t[i + 1] - T.t[i]);
        // (sqrt(2 A ds + V^2) - V) / A of :88, rationalised: same value, no cancellation (in fp32 the original loses 3 digits at |A| ~ 1e-3)
        real dt = fabs(Ai) < real(1e-3) ? ds / T.V[i] : real(2.0) * ds / (sqrt(real(2.0) * Ai * ds + T.V[i] * T.V[i]) + T.V[i]);
        o[0] = T.s[i] + ds; o[1] = sqrt(best) * sgn(cr); o[2] = T.t[i] + dt; o[3] = (real)i;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// nodes record per node: q[6], u[2] (physical units), V, kappa  -> 10 doubles
struct NodeRec { real q0, q1, q2, q3, q4, q5, u0, u1, pV, pK; };
// Launch order of the solve (and, in the chunked step, of update_QP): the nodes kernels file every instance either at the FRONT of `order` (likely slow) or from the
// BACK (likely quick).  k_solve ends with its slowest wave and a slow instance costs 5-10 quick ones, so the slow ones should start first.  A hint: it moves no result.
//  * warm instance: slow if the previous step needed the interior point or more than one polish round (in closed loop ~1 % of the batch, mostly the same ones);
//  * cold instance: slow if (i) the steering has far to go at the rate limit -- |delta(node 1) - delta(now)| / deltadot_max > 1.5 x the short horizon: a rate-limited
//    ramp, one or two rows join the working set per round -- or (ii) it starts outside or at the edge of the stability envelope (margin of (Uy, r) below 0.05:
//    soft rows active from the first stage on).  These two groups hold nearly all instances the active-set guess does not serve on the reference's paths.
struct OrderOut { const int* prev_status; const int* prev_iters; const int* prev_polish; int* order; int* cnt; int* slow; };      // slow: the verdict per instance, [B]
PG_DEV void file_order(const OrderOut& F, int B, int b, bool slow) {
    if (!F.order) return;
    const int pos = slow ? atomicAdd(F.cnt, 1) : B - 1 - atomicAdd(F.cnt + 1, 1);
    F.order[pos] = b;
    F.slow[b] = slow ? 1 : 0;
}
// with the safety row installed the order is filed AGAIN once (M, b) are known: an instance whose row is violated at the current control goes through the interior
// point (k_solve skips the empty-set rounds there), i.e. it is slow whatever the nodes kernel thought of it
__global__ __launch_bounds__(256) void k_order_hji(DevCfg C, int B, const real* __restrict__ control, const real* __restrict__ Mb, OrderOut F) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const real* u = control + (size_t)b * 3; const real* m = Mb + (size_t)b * 4;
    const bool hot = m[0] * (u[0] / C.un0) + m[1] * ((u[1] + u[2]) / C.un1) + m[2] < real(0.0);
    const bool slow = F.slow[b] != 0 || hot;
    const int pos = slow ? atomicAdd(F.cnt, 1) : B - 1 - atomicAdd(F.cnt + 1, 1);
    F.order[pos] = b;
}
PG_DEV bool warm_slow(const OrderOut& F, int b) { return F.order && (F.prev_status[b] != PG_SOLVED || F.prev_iters[b] > 0 || F.prev_polish[b] != 1); }
PG_DEV void put_node(real* __restrict__ ND, int i, const NodeRec& r) {
    real* o = ND + i * 10;
    o[0] = r.q0; o[1] = r.q1; o[2] = r.q2; o[3] = r.q3; o[4] = r.q4; o[5] = r.q5; o[6] = r.u0; o[7] = r.u1; o[8] = r.pV; o[9] = r.pK;
}
#ifdef PG_TIMELINE        // (diagnostic build: wall-clock marks per block of the pipelined launch -- slots 0 entry, 1 end of the wait for the nodes, 2 exit, 3 interval (1000: a nodes block, whose slots 4..7 are: trajectory staged, measured state seeded, node 1, node 3) -- read by pg_debug_pipeline_timeline)
__device__ unsigned long long g_nl_timeline[8 * 8192];
#define PG_NL_MARK(slot, v) do { if (threadIdx.x == 0 && blockIdx.x < 8192) g_nl_timeline[8 * blockIdx.x + (slot)] = (v); } while (0)
#else
#define PG_NL_MARK(slot, v) do { } while (0)
#endif
// the two searched channels (t, s) of the shared trajectory into LDS.  Eight strides per trip, every load of a trip issued before the first store: rolled one stride per
// trip, each of the 16 trips at L = 1000 was a round trip to the L2 -- and this copy is the first thing on the serial chain of the seeding recurrence.
PG_DEV void stage_trajectory(const TrajView& T, real* __restrict__ sh) {
    const int Lt = T.L, bd = (int)blockDim.x;
    for (int i0 = (int)threadIdx.x; i0 < Lt; i0 += 8 * bd) {
        real vt[8], vs[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const int i = i0 + u * bd; const int ic = i < Lt ? i : Lt - 1; vt[u] = T.t[ic]; vs[u] = T.s[ic]; }
#pragma unroll
        for (int u = 0; u < 8; u++) { const int i = i0 + u * bd; if (i < Lt) { sh[i] = vt[u]; sh[Lt + i] = vs[u]; } }
    }
}
// PUB (k_nodes_linearize, the pipelined nodes + update_QP launch): a wavefront whose 64 instances are all cold publishes, after every node of the seeding
// recurrence, how many nodes of its instances are complete (`progress[blk]`, release at device scope) -- the linearisation of interval t only needs nodes t, t + 1.
// The two scalar recurrences of the cold seeding (coupled_lat_long.jl:117-141), in functions of their own with contraction off: they are evaluated in several places (the serial
// loop, the chain that runs ahead, the re-start behind a node whose limit bound) and in two instantiations of nodes_body, and every one of them must round alike.
PG_DEV void advance_vs(real& V, real& s, real A, real tau) {
#pragma clang fp contract(off)
    V = V + A * tau;
    s = s + V * tau + A * tau * tau * real(0.5);
}
PG_DEV real commanded_accel(const DevControl& cp, real trajA, real trajV, real V, real ds, real tau, bool traj_mode) {
#pragma clang fp contract(off)
    real A_des = trajA + cp.k_V * (trajV - V) / tau + (traj_mode ? -cp.k_s * ds / tau / tau : real(0.0));
    return jmin(jmax(A_des, (cp.V_min - V) / tau), (cp.V_max - V) / tau);
}
// What the cold seeding derives from the MEASURED state before its recurrence starts (coupled_lat_long.jl:103-119): speed along the path, sines and cosines of the side-slip
// and steering angles, the front lateral tire force, the acceleration of the first step from the full nonlinear model.  The launch-per-phase kernel (k_nodes) and the pipelined
// one (k_nodes_linearize) each inline their own copy of nodes_body, and the two launch shapes must give the same bits (tests/test_gpu_full_size.py, test_gpu_api_contract.py).
// Everything else on the seeding path rounds alike by construction (contraction off); the tire-force code has no such pragma (the linearisation wants its multiply-adds
// fused), and its contraction came out differently in the two copies as soon as the code around it changed -- 14 of 2560 engine-limited instances one ulp apart in Fx of their
// short nodes.  Its inputs and results therefore pass through opaque register moves: the block is compiled the same way whatever surrounds it.  (A function of its own that is
// NOT inlined does the same by construction -- and its call frame, 800 B of scratch in a kernel that had none, cost the pipelined launch 0.32 -> 0.365 ms.)
struct MeasuredSeed { real V, beta0, sb0, cb0, sd0, cd0, Fyf0, A1; };
PG_DEV void seed_from_measured_state(const DevVehicle& P, real dpsi, real Ux0, real Uy0, real r0, real d0, real Fxf0, real Fxr0, MeasuredSeed& o) {
    asm volatile("" : "+v"(dpsi), "+v"(Ux0), "+v"(Uy0), "+v"(r0), "+v"(d0), "+v"(Fxf0), "+v"(Fxr0));
    real sdp, cdp; pg_sincos(dpsi, &sdp, &cdp);
    o.V = Ux0 * cdp - Uy0 * sdp;
    o.beta0 = atan2(Uy0, Ux0);
    pg_sincos(o.beta0, &o.sb0, &o.cb0); pg_sincos(d0, &o.sd0, &o.cd0);
    {   // lateral_tire_forces(bicycle, q0, u0): raw (delta, Fxf, Fxr), no actuator limits (:110; vehicle_dynamics.jl:78-87)
        real sd = o.sd0, cd = o.cd0, Fyr0;
        real af = tan(atan2(Uy0 + P.a * r0, Ux0) - d0), ar = tan(atan2(Uy0 - P.b * r0, Ux0));
        lateral_forces<real>(P, af, ar, Fxf0, Fxr0, sd, cd, o.Fyf0, Fyr0);
    }
    real dUx, dUy, dr;                      // i == 1 of the reference loop: acceleration from the full nonlinear model (:117-119)
    world_body_rhs<real>(P, Ux0, Uy0, r0, d0, Fxf0 + Fxr0, dUx, dUy, dr);
    o.A1 = (dUx - r0 * Uy0) * cdp - (dUy + r0 * Ux0) * sdp;
    asm volatile("" : "+v"(o.V), "+v"(o.beta0), "+v"(o.sb0), "+v"(o.cb0), "+v"(o.sd0), "+v"(o.cd0), "+v"(o.Fyf0), "+v"(o.A1));
}
// Lanes per instance in the nodes kernels (round 5).  The cold seeding couples node i + 1 to node i through ONE number: the acceleration A the steady-state solve of node i
// returns.  On the long horizon that is the commanded acceleration, pulled back onto the friction circle (limit_accel), to rounding -- unless an actuator or tire limit binds
// INSIDE the solve (measured on four of the reference's paths, 256 cold instances each: 0 of 19,456 long nodes differ by more than 2e-16 relative; EXPERIMENTS.md 11).  So the
// PG_NODES_LPN lanes of an instance run (V, s) ahead on that value for PG_NODES_LPN nodes (a cheap chain: two searches and a square root per node), each lane then evaluates the
// four-iteration solve of ITS node -- side by side instead of one after the other: the solves are 5/6 of the serial chain --, and every node whose solve returns the acceleration
// assumed (to 1e-12) is committed; a node where a limit did bind is committed with the solve's value and the nodes behind it are recomputed from there (>= 1 node per pass).
// The short nodes (one iteration from the measured state: not a fixed point) stay serial, evaluated redundantly by the lanes of the instance.
#ifndef PG_NODES_LPN
#define PG_NODES_LPN 2
#endif
constexpr int NODES_LPN = PG_NODES_LPN, NODES_IPB = 64 / NODES_LPN;      // lanes per instance, instances per 64-lane block of a nodes kernel
template <bool STAGED, bool PUB> PG_DEV void nodes_body(const DevCfg& C, int B, int blk, const real* __restrict__ state, const real* __restrict__ control, const tdouble* __restrict__ toff,
                        const int* __restrict__ solved, const real* __restrict__ sep, const tdouble* __restrict__ ts, const tdouble* __restrict__ dt,
                        const tdouble* __restrict__ prev_ts, const real* __restrict__ prev_x, real* __restrict__ nodes, OrderOut F, real* __restrict__ naux, int* __restrict__ progress, unsigned long long pub_mask) {
    // the two searched channels (t, s) are staged in LDS when they fit: every node costs three binary searches whose ~10 dependent probes each
    // would otherwise pay L2 latency (the kernel is a 64-wave serial recurrence: latency, not bandwidth, is its whole cost)
    extern __shared__ real sh_traj[];
    TrajView T = C.traj;
    if constexpr (STAGED) {                                 // single shared trajectory; compile-time so the searches compile to ds_read, not flat loads
        stage_trajectory(C.traj, sh_traj);
        __syncthreads();
        T.t = sh_traj; T.s = sh_traj + T.L;
    }
    if constexpr (PUB) PG_NL_MARK(4, wall_clock64());
    const int b = blk * NODES_IPB + (int)threadIdx.x / NODES_LPN, lp = (int)threadIdx.x % NODES_LPN;      // lp: this lane's place among the lanes of its instance
    if (b >= B) return;
    const bool lead = lp == 0;                            // (the lane that writes what all lanes of the instance compute alike)
    bool publish = false;
    if constexpr (PUB) publish = __all(solved[b] == 0) != 0;          // (over the live lanes of the wavefront; lane 0 is always live)
    if constexpr (!STAGED) T = traj_of(C, b);
    const DevVehicle& P = C.veh;
    const real* q0 = state + (size_t)b * 6; const real* u0 = control + (size_t)b * 3;
    const tdouble* TS = ts + (size_t)b * C.NN; const tdouble* DT = dt + (size_t)b * C.N;
    real* ND = nodes + (size_t)b * C.NN * 10;
    const real s0 = sep[(size_t)b * 4], e0 = sep[(size_t)b * 4 + 1];
    const real psi0 = q0[2], Ux0 = q0[3], Uy0 = q0[4], r0 = q0[5];
    const real d0 = u0[0], Fxf0 = u0[1], Fxr0 = u0[2];
    TrajS tj = traj_at_s(T, s0);                                                   // :76
    const real ds0 = s0 - traj_s_at_time(T, TS[0]);                              // :77
    const real dpsi = adiff(psi0, tj.psi);                                       // :78
    // node 1 of the reference (index 0 here) is the measured state in both branches (:79-85, and i == 1 of the cold loop)
    NodeRec r;
    r.q0 = ds0; r.q1 = Ux0; r.q2 = Uy0; r.q3 = r0; r.q4 = dpsi; r.q5 = e0; r.u0 = d0; r.u1 = Fxf0 + Fxr0; r.pV = tj.V; r.pK = tj.kappa;
    if (lead) put_node(ND, 0, r);
    if (solved[b]) {                                                               // :82-102 with update_interpolations! (:189-195)
        const tdouble* PT = prev_ts + (size_t)b * C.NN; const real* PX = prev_x + (size_t)b * C.NN * 8;
        const real tlast = PT[C.NN - 1];
        for (int i = 1 + lp; i < C.NN; i += NODES_LPN) {       // (the nodes of a warm instance are independent of each other: one in NODES_LPN per lane)
            real t = TS[i];
            real tq = (t < tlast) ? t : tlast;
            int j = clampi(count_leq(PT, C.NN, tq), 1, C.NN - 1) - 1;
            real w = (tq - PT[j]) / (PT[j + 1] - PT[j]);
            const real* a = PX + j * 8; const real* c = a + 8;
            r.q0 = (real(1.0) - w) * a[0] + w * c[0]; r.q1 = (real(1.0) - w) * a[1] + w * c[1]; r.q2 = (real(1.0) - w) * a[2] + w * c[2];
            r.q3 = (real(1.0) - w) * a[3] + w * c[3]; r.q4 = (real(1.0) - w) * a[4] + w * c[4]; r.q5 = (real(1.0) - w) * a[5] + w * c[5];
            r.u0 = ((real(1.0) - w) * a[6] + w * c[6]) * C.un0; r.u1 = ((real(1.0) - w) * a[7] + w * c[7]) * C.un1;
            real s = traj_s_at_time(T, t) + r.q0;                                // :96
            tj = traj_at_s(T, s);
            r.pV = tj.V; r.pK = tj.kappa;
            put_node(ND, i, r);
        }
        for (int i = lp; i < C.NN; i += NODES_LPN) naux[((size_t)b * C.NN + i) * 4] = NAN;          // (nothing deferred for a warm instance: k_nodes_angles skips it)
        if (lead) file_order(F, B, b, warm_slow(F, b));
        return;
    }
    if (lead) naux[(size_t)b * C.NN * 4] = NAN;           // node 0 is the measured state
    // cold start :103-141
    MeasuredSeed ms; seed_from_measured_state(P, dpsi, Ux0, Uy0, r0, d0, Fxf0, Fxr0, ms);
    real V = ms.V;
    const real beta0 = ms.beta0, sb0 = ms.sb0, cb0 = ms.cb0, sd0 = ms.sd0, cd0 = ms.cd0, Fyf0 = ms.Fyf0;
    const bool traj_mode = !(toff[b] != toff[b]);
    // i == 1 of the reference loop: acceleration from the full nonlinear model (:117-119); its node record is already written above
    real s = s0, d1 = d0, Fx1 = real(0.0);
    advance_vs(V, s, ms.A1, DT[0]);
    if constexpr (PUB) PG_NL_MARK(5, wall_clock64());
    const bool ahead = NODES_LPN > 1 && C.Ns + 1 < C.NN;     // the long nodes NODES_LPN at a time (below); the loop here then ends with the short horizon
    const int i_serial_end = ahead ? C.Ns + 1 : C.NN;
#pragma unroll 1
    for (int i = 1; i < i_serial_end; i++) {
        real tau = (i == C.NN - 1) ? DT[i - 1] : DT[i];
        real s_ref; traj_lookup2(T, s, TS[i], tj, s_ref);
        real ds = s - s_ref;
        const real A_des = commanded_accel(C.cp, tj.A, tj.V, V, ds, tau, traj_mode);
        const bool shortp = i <= C.Ns;
        // :122 short nodes: one iteration from the measured (r0, beta0, delta0, Fyf0); :128 long nodes: four iterations from (V kappa, 0, 0, 0)
        Steady est = steady_state(P, V, A_des, tj.kappa, shortp ? 1 : 4, shortp ? r0 : V * tj.kappa, shortp ? beta0 : real(0.0), shortp ? sb0 : real(0.0), shortp ? cb0 : real(1.0),
                                  shortp ? d0 : real(0.0), shortp ? sd0 : real(0.0), shortp ? cd0 : real(1.0), shortp ? Fyf0 : real(0.0), true);
        r.q0 = ds;
        r.q1 = shortp ? Ux0 : est.Ux; r.q2 = shortp ? Uy0 : est.Uy; r.q3 = shortp ? r0 : est.r;
        r.q4 = shortp ? adiff(psi0, tj.psi) : real(0.0); r.q5 = shortp ? e0 : real(0.0);      // long nodes: q4 = -beta and u0 = delta are finished by k_nodes_angles
        r.u0 = real(0.0); r.u1 = est.Fx; r.pV = tj.V; r.pK = tj.kappa;
        if (lead) {
            put_node(ND, i, r);
            real* ax = naux + ((size_t)b * C.NN + i) * 4;
            ax[0] = est.ang_y; ax[1] = est.ang_x; ax[2] = est.ang_t; ax[3] = (!shortp && est.beta_is_tan) ? est.tb : NAN;      // (NaN: q4 stands as written)
        }
        if (i == 1) { d1 = atan2(est.ang_y, est.ang_x) - atan(est.ang_t); Fx1 = est.Fx; }          // (the launch-order hint needs this one angle now)
        if constexpr (PUB) { if (i == 1) PG_NL_MARK(6, wall_clock64()); if (i == 3) PG_NL_MARK(7, wall_clock64()); }
        if constexpr (PUB) {
            if (publish && ((pub_mask >> i) & 1ull)) {                 // nodes 0..i of all 64 instances are in memory: release, then the count
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");      // (a RELEASE: write-back only -- __threadfence() also invalidates this wavefront's caches, which hold the trajectory it keeps searching)
                if (threadIdx.x == 0) __hip_atomic_store(progress + blk, i + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        advance_vs(V, s, est.A, tau);
    }
    if (ahead) {
        const real tolA = sizeof(real) == 8 ? real(1e-12) : real(4e-6);      // (fp32: the limit and the solve's own value differ by a few ulp as a rule; tests/test_gpu_f32.py holds the nodes against the serial form, option "nodes_serial")
        const int base = (int)(threadIdx.x & 63u) - lp;        // first lane of this instance in the wavefront
        int i = C.Ns + 1, done = C.Ns + 1;                     // i: next node of this instance = its number of complete nodes; done: the wavefront's (min over its instances)
#pragma unroll 1
        while (__any(i < C.NN)) {
            // (V, s) ahead on the commanded acceleration: every lane of the instance runs the same cheap chain and keeps the arguments of ITS node
            real Vs = V, ss = s, Vm = V, sm = s, Adm = real(0.0), kpm = real(0.0), taum = real(1.0), dsm = real(0.0), pVm = real(0.0), Agm = real(0.0);
            real Agj[NODES_LPN], Vst[NODES_LPN], sst[NODES_LPN], tauj[NODES_LPN];
#pragma unroll
            for (int j = 0; j < NODES_LPN; j++) {
                const int ii = i + j < C.NN ? i + j : C.NN - 1;
                const real tau = (ii == C.NN - 1) ? DT[ii - 1] : DT[ii];
                TrajS tjj; real s_ref; traj_lookup2(T, ss, TS[ii], tjj, s_ref);
                const real ds = ss - s_ref;
                const real A_des = commanded_accel(C.cp, tjj.A, tjj.V, Vs, ds, tau, traj_mode);
                real At = A_des, Ar; limit_accel(P, Vs, tjj.kappa, At, Ar);
                if (j == lp) { Vm = Vs; sm = ss; Adm = A_des; kpm = tjj.kappa; taum = tau; dsm = ds; pVm = tjj.V; Agm = At; }
                Vst[j] = Vs; sst[j] = ss; tauj[j] = tau; Agj[j] = At;
                advance_vs(Vs, ss, At, tau);
            }
            // this lane's node: the four-iteration solve from (V kappa, 0, 0, 0) (:128), exactly as the serial form evaluates it
            const Steady est = steady_state(P, Vm, Adm, kpm, 4, Vm * kpm, real(0.0), real(0.0), real(1.0), real(0.0), real(0.0), real(1.0), real(0.0), true);
            const bool match = fabs(est.A - Agm) <= tolA * (real(1.0) + fabs(Agm));       // (a NaN never matches: the solve's own value goes on, as in the serial form)
            const unsigned long long mb = __ballot(match);
            // nodes i .. i + c - 1 are valid: node i always, node i + j while every node before it returned the acceleration assumed
            int c = 1;
#pragma unroll
            for (int j = 1; j < NODES_LPN; j++) c += (c == j && ((mb >> (base + j - 1)) & 1ull) && i + j < C.NN && !C.nodes_serial) ? 1 : 0;
            if (i < C.NN && lp < c) {
                NodeRec rr;
                rr.q0 = dsm; rr.q1 = est.Ux; rr.q2 = est.Uy; rr.q3 = est.r; rr.q4 = real(0.0); rr.q5 = real(0.0);      // q4 = -beta and u0 = delta are finished by k_nodes_angles
                rr.u0 = real(0.0); rr.u1 = est.Fx; rr.pV = pVm; rr.pK = kpm;
                put_node(ND, i + lp, rr);
                real* ax = naux + ((size_t)b * C.NN + i + lp) * 4;
                ax[0] = est.ang_y; ax[1] = est.ang_x; ax[2] = est.ang_t; ax[3] = est.beta_is_tan ? est.tb : NAN;
            }
            // state behind the last valid node: ALWAYS from the acceleration its own solve returned (every lane of the instance alike), as the serial form advances -- round 5
            // kept the chain's value where that node had matched, i.e. the ASSUMED acceleration, up to 1e-12 (4e-6 in fp32) away per pass
            if (i < C.NN) {
                const int jl = c - 1;
                const real A_last = __shfl(est.A, base + jl);
                real Vs0 = Vst[0], ss0 = sst[0], tl = tauj[0];
#pragma unroll
                for (int j = 1; j < NODES_LPN; j++) if (j == jl) { Vs0 = Vst[j]; ss0 = sst[j]; tl = tauj[j]; }
                real Vx = Vs0, sx = ss0; advance_vs(Vx, sx, A_last, tl);
                V = Vx; s = sx;
                i += c;
            }
            if constexpr (PUB) {
                const int prev = done;
                while (done < C.NN && __all(i > done)) done++;
                if (publish && done > prev) {      // nodes prev .. done - 1 of every instance of the wavefront are in memory: publish when one of them is a publication point
                    const unsigned long long lo_new = done >= 64 ? ~0ull : ((1ull << done) - 1ull), lo_old = (1ull << prev) - 1ull;
                    if (pub_mask & 0x7FFFFFFFFFFFFFFFull & lo_new & ~lo_old) {
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                        if (threadIdx.x == 0) __hip_atomic_store(progress + blk, done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
        }
    }
    if (F.order && lead) {
        bool slow = fabs(d1 - d0) > real(1.5) * C.cp.deltadot_max * ((real)C.Ns * (real)C.dt_short);
        const Envelope e = stable_limits(P, Ux0, Fx1 > real(0.0) ? Fx1 * P.fwd_frac : Fx1 * P.fwb_frac, Fx1 > real(0.0) ? Fx1 * P.rwd_frac : Fx1 * P.rwb_frac);
#pragma unroll
        for (int k = 0; k < 4; k++) slow = slow || (e.H[k][0] * Uy0 + e.H[k][1] * r0 - e.G[k] > real(-0.05));
        file_order(F, B, b, slow);
    }
}

template <bool STAGED> __global__ __launch_bounds__(64) void k_nodes(DevCfg C, int B, const real* __restrict__ state, const real* __restrict__ control, const tdouble* __restrict__ toff,
                        const int* __restrict__ solved, const real* __restrict__ sep, const tdouble* __restrict__ ts, const tdouble* __restrict__ dt,
                        const tdouble* __restrict__ prev_ts, const real* __restrict__ prev_x, real* __restrict__ nodes, OrderOut F, real* __restrict__ naux) {
    nodes_body<STAGED, false>(C, B, (int)blockIdx.x, state, control, toff, solved, sep, ts, dt, prev_ts, prev_x, nodes, F, naux, nullptr, 0ull);
}

// The angles k_nodes deferred (steady_state(.., defer = true)): delta = atan2(y, x) - atan(t) of every seeded node and beta = atan(tb) of the long ones -- the same
// expressions on the same arguments, evaluated with lane = (instance, node) instead of inside the serial chain of the instance (three inverse tangents per node were a
// fifth of that chain's instructions).
__global__ __launch_bounds__(256) void k_nodes_angles(DevCfg C, int B, const real* __restrict__ naux, real* __restrict__ nodes, const int* __restrict__ only_if = nullptr) {
    if (only_if && *only_if == 0) return;                         // (repair launch behind k_nodes_linearize: runs only when a waiting wavefront of that launch gave up)
    // (grid-stride: the ordinary launch has one thread per node; the repair launch is a small grid -- an empty launch then costs ~2 us instead of 4.6)
    for (long gid = (long)blockIdx.x * blockDim.x + threadIdx.x; gid < (long)B * C.NN; gid += (long)gridDim.x * blockDim.x) {
        const real* ax = naux + (size_t)gid * 4;
        if (ax[0] != ax[0]) continue;                             // measured node / warm instance: nothing deferred
        real* nd = nodes + (size_t)gid * 10;
        nd[6] = atan2(ax[0], ax[1]) - atan(ax[2]);
        if (ax[3] == ax[3]) nd[4] = -atan(ax[3]);
    }
}
// Warm branch of compute_linearization_nodes! (coupled_lat_long.jl:82-102) for a batch in which EVERY instance has a previous solution (closed loop after the
// first step): the 31 nodes of an instance are independent of each other there -- node i interpolates the previous solution at ts[i] and looks the reference up at
// the resulting arclength -- so the lane is (instance, node) instead of the instance (k_nodes runs the same arithmetic node after node in one lane because the
// COLD seeding is a recurrence; mixed batches keep using it).  Same operations per node as k_nodes' warm branch: bit-identical nodes.
template <bool STAGED> __global__ __launch_bounds__(256) void k_nodes_warm(DevCfg C, int B, const real* __restrict__ state, const real* __restrict__ control, const real* __restrict__ sep,
                             const tdouble* __restrict__ ts, const tdouble* __restrict__ prev_ts, const real* __restrict__ prev_x, real* __restrict__ nodes, OrderOut F) {
    extern __shared__ real sh_traj[];
    TrajView T = C.traj;
    if constexpr (STAGED) {
        stage_trajectory(C.traj, sh_traj);
        __syncthreads();
        T.t = sh_traj; T.s = sh_traj + T.L;
    }
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int b = (int)(gid / C.NN), i = (int)(gid - (long)b * C.NN);
    if (b >= B) return;
    if constexpr (!STAGED) T = traj_of(C, b);
    const tdouble* TS = ts + (size_t)b * C.NN;
    real* ND = nodes + (size_t)b * C.NN * 10;
    NodeRec r;
    if (i == 0) {                                                                   // node 1 of the reference: the measured state (:76-85)
        const real* q0 = state + (size_t)b * 6; const real* u0 = control + (size_t)b * 3;
        const real s0 = sep[(size_t)b * 4], e0 = sep[(size_t)b * 4 + 1];
        TrajS tj = traj_at_s(T, s0);
        r.q0 = s0 - traj_s_at_time(T, TS[0]); r.q1 = q0[3]; r.q2 = q0[4]; r.q3 = q0[5]; r.q4 = adiff(q0[2], tj.psi); r.q5 = e0;
        r.u0 = u0[0]; r.u1 = u0[1] + u0[2]; r.pV = tj.V; r.pK = tj.kappa;
        file_order(F, B, b, warm_slow(F, b));
    } else {                                                                        // :87-101 with update_interpolations! (:189-195)
        const tdouble* PT = prev_ts + (size_t)b * C.NN; const real* PX = prev_x + (size_t)b * C.NN * 8;
        const real tlast = PT[C.NN - 1];
        real t = TS[i];
        real tq = (t < tlast) ? t : tlast;
        int j = clampi(count_leq(PT, C.NN, tq), 1, C.NN - 1) - 1;
        real w = (tq - PT[j]) / (PT[j + 1] - PT[j]);
        const real* a = PX + j * 8; const real* c = a + 8;
        r.q0 = (real(1.0) - w) * a[0] + w * c[0]; r.q1 = (real(1.0) - w) * a[1] + w * c[1]; r.q2 = (real(1.0) - w) * a[2] + w * c[2];
        r.q3 = (real(1.0) - w) * a[3] + w * c[3]; r.q4 = (real(1.0) - w) * a[4] + w * c[4]; r.q5 = (real(1.0) - w) * a[5] + w * c[5];
        r.u0 = ((real(1.0) - w) * a[6] + w * c[6]) * C.un0; r.u1 = ((real(1.0) - w) * a[7] + w * c[7]) * C.un1;
        real s = traj_s_at_time(T, t) + r.q0;                                   // :96
        TrajS tj = traj_at_s(T, s);
        r.pV = tj.V; r.pK = tj.kappa;
    }
    put_node(ND, i, r);
}

// ------------------------------------------------------------------------------------------------------------------
// update_QP! of the coupled formulation (coupled_lat_long.jl:315-368) in ONE kernel: `linearize` of every interval + c + u-normalisation (:335-353), the
// stability envelope and the bounds (:354-367), q_curr/u_curr (:332-333) and the safety row (:345-346).
// linearize = RK4 (nsub sub-steps) of VehicleModel{TrackingBicycleModel} on forward-mode numbers (third-party LinearDynamicsModels; restated in EXPERIMENTS.md 2).
// TWO lanes per (instance, interval), four tangent directions each (DK<4>): lane 0 carries d/d(Ux, Uy, r, dpsi), lane 1 d/d(u0[0], u0[1], uf[0], uf[1]).
// The tracking model does not read ds or e (vehicle_dynamics.jl:159-183: neither appears on the right-hand side), so dPhi/d(ds) = e_0 and dPhi/d(e) = e_5
// EXACTLY -- those two columns of A are written as constants instead of being integrated (round 1 propagated all ten tangents in five lanes of two, i.e.
// five primal trajectories per interval; now two).  c = Phi - A q - B0 u0 - Bf uf is finished in the same lanes (one shuffle inside the lane pair), lane 1
// also evaluates stable_limits for the interval.
// (the lane pair (g = 0, 1) of an interval must be two adjacent lanes of one wavefront; a lane that is not `live` computes along with its partner and stores nothing)
// K tangent directions per lane, G = 8 / K adjacent lanes per interval (K = 4: the lane PAIR of the description above; K = 2, 1: four / eight lanes per interval for
// small batches, where the kernel is a handful of wavefronts and its duration is the latency of ONE lane -- fewer directions per lane, shorter chain).  Direction
// j = g K + d: j < 4 the state components (Ux, Uy, r, dpsi), j = 4, 5: u0, j = 6, 7: uf.  Every direction is propagated by the same arithmetic whatever K is, so the
// Jacobians are bit-identical across K; c differs by the order of its cross-lane sum (1e-16).
// ND = 6: an interval of the short horizon (zero-order hold: uf is not a variable, directions 6 and 7 vanish identically) with the six remaining directions on
// G = 6 / K lanes; the Bf block is written as zeros and c sums the same six products in the same order -- the same bits as ND = 8 gives on such an interval.
// (n0, n1: the node records of the interval's two ends -- in the nodes array, or a copy of them: k_nodes_linearize)
template <int K, int ND = 8>
PG_DEV void linearize_lanes_at(const DevCfg& C, int b, int t, int g, bool live, const real* __restrict__ n0, const real* __restrict__ n1, const tdouble* __restrict__ dt,
                               const real* __restrict__ hji_Mb, real* __restrict__ qp) {
    constexpr int G = ND / K;
    const bool ramp = ND == 8 && t >= C.Ns;
    const real h_total = dt[(size_t)b * C.N + t];
    // value of the state and its K tangent directions (direction j = g K + d: j < 4 seeds component j + 1; 4, 5: u0; 6, 7: uf)
    struct XD { real v, d[K]; };
    XD x[6];
#pragma unroll
    for (int k = 0; k < 6; k++) {
        x[k].v = n0[k];
#pragma unroll
        for (int d = 0; d < K; d++) x[k].d[d] = (g * K + d < 4 && g * K + d + 1 == k) ? real(1.0) : real(0.0);
    }
    const real u0a = n0[6], u0b = n0[7], ufa = ramp ? n1[6] : n0[6], ufb = ramp ? n1[7] : n0[7];
    // tangents of the interpolated control u(tau) = u0 + (uf - u0) w: direction 4 / 5 carries (1 - w) on delta / Fx, direction 6 / 7 carries w
    real s0a[K], s0b[K], sfa[K], sfb[K];
#pragma unroll
    for (int d = 0; d < K; d++) {
        const int j = g * K + d;
        s0a[d] = j == 4 ? real(1.0) : real(0.0); s0b[d] = j == 5 ? real(1.0) : real(0.0); sfa[d] = j == 6 ? real(1.0) : real(0.0); sfb[d] = j == 7 ? real(1.0) : real(0.0);
    }
    const real pV0 = n0[8], pK0 = n0[9], pV1 = ramp ? n1[8] : n0[8], pK1 = ramp ? n1[9] : n0[9];
    const int nsub = C.nsub; const real h = h_total / nsub;
    const real h6 = h / real(6.0);
#pragma unroll 1
    for (int i = 0; i < nsub; i++) {
        const real t0 = i * h;
        XD xx[6], acc[6];
#pragma unroll
        for (int k = 0; k < 6; k++) {
            xx[k] = x[k]; acc[k].v = real(0.0);
#pragma unroll
            for (int d = 0; d < K; d++) acc[k].d[d] = real(0.0);
        }
        // classical RK4 as one rolled stage loop.  Per stage: the right-hand side and its local Jacobian ONCE (scalars: tracking_jac), then every direction is a handful
        // of multiply-adds -- k.d = J (xx.d, du) -- in an order that does not depend on K: the Jacobians are bit-identical whatever the lane arrangement
#pragma unroll 1
        for (int st = 0; st < 4; st++) {
            const real cst_ = st == 0 ? real(0.0) : (st == 3 ? real(1.0) : real(0.5));        // stage time fraction
            const real wgt = (st == 0 || st == 3) ? real(1.0) : real(2.0);              // quadrature weight (x h/6)
            const real nxt = (st == 2 ? real(1.0) : real(0.5)) * h;                     // coefficient of k in the NEXT stage's evaluation point
            const real w = ramp ? (t0 + cst_ * h) / h_total : real(0.0);
            const real ua = u0a + (ufa - u0a) * w, ub = u0b + (ufb - u0b) * w;
            real qv[6];
#pragma unroll
            for (int k = 0; k < 6; k++) qv[k] = xx[k].v;
            TrackJac J;
            tracking_jac(C.veh, qv, ua, ub, pV0 + (pV1 - pV0) * w, pK0 + (pK1 - pK0) * w, J);
#pragma unroll
            for (int k = 0; k < 6; k++) { acc[k].v = acc[k].v + J.f[k] * wgt; xx[k].v = x[k].v + J.f[k] * nxt; }
#pragma unroll
            for (int d = 0; d < K; d++) {
                const real tU = xx[1].d[d], tY = xx[2].d[d], tR = xx[3].d[d], tP = xx[4].d[d];
                const real da = s0a[d] + (sfa[d] - s0a[d]) * w, db = s0b[d] + (sfb[d] - s0b[d]) * w;
                real kd[6];
                kd[0] = J.a0[0] * tU + J.a0[1] * tY + J.a0[2] * tP;
                // One lane per interval (K == ND): the direction index is a compile-time number once this loop is unrolled, and the input terms exist for two directions each
                // (da for j = 4, 6; db for j = 5, 7) -- the others would add 0 x J: the same value without them (36 of 288 multiply-adds per stage)
                const int j = g * K + d;
                const bool has_a = K != ND || j == 4 || j == 6, has_b = K != ND || j == 5 || j == 7;
#pragma unroll
                for (int m = 0; m < 3; m++) {
                    real v = J.b[m][0] * tU + J.b[m][1] * tY + J.b[m][2] * tR;
                    if (has_a) v = v + J.b[m][3] * da;
                    if (has_b) v = v + J.b[m][4] * db;
                    kd[1 + m] = v;
                }
                kd[4] = tR + J.a4[0] * tU + J.a4[1] * tY + J.a4[2] * tP;
                kd[5] = J.a5[0] * tU + J.a5[1] * tY + J.a5[2] * tP;
                // (the tangents of ds and e feed nothing back -- the model reads neither --: pure quadratures, summed into the state as they come instead of through an
                //  accumulator of their own: 2 K fewer live values)
                x[0].d[d] = x[0].d[d] + kd[0] * (wgt * h6); x[5].d[d] = x[5].d[d] + kd[5] * (wgt * h6);
#pragma unroll
                for (int k = 1; k < 5; k++) { acc[k].d[d] = acc[k].d[d] + kd[k] * wgt; xx[k].d[d] = x[k].d[d] + kd[k] * nxt; }
            }
        }
#pragma unroll
        for (int k = 0; k < 6; k++) {
            x[k].v = x[k].v + acc[k].v * h6;
            if (k >= 1 && k <= 4) {
#pragma unroll
                for (int d = 0; d < K; d++) x[k].d[d] = x[k].d[d] + acc[k].d[d] * h6;
            }
        }
    }
    // this lane's share of c_i = Phi_i - A_i. q - B0_i. u0 - Bf_i. uf   (raw, un-normalised Jacobians: coupled_lat_long.jl:336-353)
    // The eight products d_j coef_j are rounded one by one, brought to the group's first lane and summed THERE in one fixed order, so that c -- like the Jacobians --
    // does not depend on K: a batch gives the same bits whether it is stepped whole or in shards of another size (tests/test_gpu_multiprocess.py)
    real part[6];
    {
#pragma clang fp contract(off)
        real coef[K];
#pragma unroll
        for (int d = 0; d < K; d++) {
            const int j = g * K + d;
            coef[d] = j < 4 ? n0[j + 1] : (j == 4 ? n0[6] : (j == 5 ? n0[7] : (ramp ? (j == 6 ? n1[6] : n1[7]) : real(0.0))));
        }
        const int base = (int)(threadIdx.x & 63u) - g;
#pragma unroll
        for (int i = 0; i < 6; i++) {
            real q8[8] = {real(0.0), real(0.0), real(0.0), real(0.0), real(0.0), real(0.0), real(0.0), real(0.0)};
#pragma unroll
            for (int jg = 0; jg < G; jg++) {
#pragma unroll
                for (int d = 0; d < K; d++) { const real pr = x[i].d[d] * coef[d]; q8[jg * K + d] = __shfl(pr, base + jg); }
            }
            real ci = x[i].v - (((q8[0] + q8[1]) + q8[2]) + q8[3]);
            if (i == 0) ci -= n0[0];
            if (i == 5) ci -= n0[5];
            part[i] = ci - (((q8[4] + q8[5]) + q8[6]) + q8[7]);
        }
    }
    if (!live) return;
    const QpOff o = qp_offsets(C.N);
    real* Q = qp + (size_t)b * C.qp_len;
    real* A = Q + o.A + 36 * t; real* B0 = Q + o.B0 + 12 * t; real* Bf = Q + o.Bf + 12 * t;
#pragma unroll
    for (int d = 0; d < K; d++) {
        const int j = g * K + d;
#pragma unroll
        for (int i = 0; i < 6; i++) {
            const real v = x[i].d[d];
            if (j < 4) A[6 * i + 1 + j] = v;
            else if (j < 6) B0[2 * i + (j - 4)] = v * (j == 4 ? C.un0 : C.un1);                          // :338,350-351 (B scaled by u_normalization)
            else Bf[2 * i + (j - 6)] = ramp ? v * (j == 6 ? C.un0 : C.un1) : real(0.0);
        }
    }
    if (g == 0) {
#pragma unroll
        for (int i = 0; i < 6; i++) { A[6 * i] = i == 0 ? real(1.0) : real(0.0); A[6 * i + 5] = i == 5 ? real(1.0) : real(0.0); Q[o.c + 6 * t + i] = part[i]; }
        if (t == 0) {
#pragma unroll
            for (int k = 0; k < 6; k++) Q[o.qcurr + k] = n0[k];
            Q[o.ucurr] = n0[6] / C.un0; Q[o.ucurr + 1] = n0[7] / C.un1;
            if (C.has_hji) { Q[o.M] = hji_Mb[(size_t)b * 4]; Q[o.M + 1] = hji_Mb[(size_t)b * 4 + 1]; Q[o.b] = hji_Mb[(size_t)b * 4 + 2]; }
            else { Q[o.M] = real(0.0); Q[o.M + 1] = real(0.0); Q[o.b] = real(1.0); }
        }
    }
    if (ND == 6 && g == G - 1) {
#pragma unroll
        for (int i = 0; i < 12; i++) Bf[i] = real(0.0);
    }
    if (g == G - 1) {
        const real Uxt = n1[1], Fx = n1[7];                                                             // :357-358
        const real Fxf = Fx > real(0.0) ? Fx * C.veh.fwd_frac : Fx * C.veh.fwb_frac, Fxr = Fx > real(0.0) ? Fx * C.veh.rwd_frac : Fx * C.veh.rwb_frac;
        const Envelope e = stable_limits(C.veh, Uxt, Fxf, Fxr);
#pragma unroll
        for (int r = 0; r < 4; r++) { Q[o.H + 8 * t + 2 * r] = e.H[r][0]; Q[o.H + 8 * t + 2 * r + 1] = e.H[r][1]; Q[o.G + 4 * t + r] = e.G[r]; }
        Q[o.dmin + t] = jmax(e.dmin, -C.veh.delta_max) / C.un0;
        Q[o.dmax + t] = jmin(e.dmax, C.veh.delta_max) / C.un0;
        Q[o.fxmax + t] = jmin(C.veh.Px_max / Uxt, C.veh.Fx_max) / C.un1;
        Q[o.ddmin + t] = -C.cp.deltadot_max * h_total / C.un0;
        Q[o.ddmax + t] = C.cp.deltadot_max * h_total / C.un0;
        Q[o.dt + t] = h_total;
    }
}
template <int K, int ND = 8>
PG_DEV void linearize_lanes(const DevCfg& C, int b, int t, int g, bool live, const real* __restrict__ nodes, const tdouble* __restrict__ dt, const real* __restrict__ hji_Mb,
                            real* __restrict__ qp) {
    const real* n0 = nodes + ((size_t)b * C.NN + t) * 10;
    linearize_lanes_at<K, ND>(C, b, t, g, live, n0, n0 + 10, dt, hji_Mb, qp);
}
PG_DEV void linearize_pair(const DevCfg& C, int b, int t, int g, bool live, const real* __restrict__ nodes, const tdouble* __restrict__ dt, const real* __restrict__ hji_Mb,
                           real* __restrict__ qp) {
    linearize_lanes<4>(C, b, t, g, live, nodes, dt, hji_Mb, qp);
}
#ifdef PG_F32
#define PG_SPLIT_WAVES 2        // fp32: the one-lane linearisation is 20 registers over two waves per SIMD; held to 256
#else
#define PG_SPLIT_WAVES 1
#endif
// the large-batch form: the Ns zero-order-hold intervals of every instance with six directions (ND = 6), the N - Ns ramp intervals with eight (ND = 8);
// blocks [0, nb_zoh) take the first group -- a wavefront runs one of the two instruction streams.  LPI = lanes per (instance, interval):
//   1 (round 4): ONE lane carries all the directions.  The scalar part of a lane (value, local Jacobian, RK4 of the state) costs as much as six directions, and a
//     lane pair pays it twice; one lane holds 468 registers (212 of them AGPRs the compiler moves through) and still issues 30 % fewer instructions per interval:
//     update_QP! of 4096 instances 0.343 -> 0.251 ms, the same bits (EXPERIMENTS.md 10.4 -- the paper argument against it, "two AGPR moves per access", was wrong).
//     fp32: 0.281 -> 0.160 ms.
//   2: the lane pair of rounds 1-3 (three / four directions per lane); batches of <= 1024 instances, where every pair is resident at once and the pair is the shorter chain.
template <int LPI>
__global__ __launch_bounds__(64, PG_SPLIT_WAVES) void k_linearize_split(DevCfg C, int B, int nb_zoh, const real* __restrict__ nodes, const tdouble* __restrict__ dt, const real* __restrict__ hji_Mb,
                                                        real* __restrict__ qp, const int* __restrict__ only_if = nullptr, int nb_total = 0) {
    if (only_if && *only_if == 0) return;                         // (repair launch behind k_nodes_linearize, see there)
    // (block-stride over the nb_total blocks of work when launched as a small repair grid; the ordinary launch -- nb_total = 0 -- has one block per block of work)
    for (int blk = (int)blockIdx.x; blk < (nb_total ? nb_total : (int)gridDim.x); blk += (int)gridDim.x) {
        const bool zoh = blk < nb_zoh;
        const int nint = zoh ? C.Ns : C.N - C.Ns;                      // intervals of this group per instance
        long gid = (long)(zoh ? blk : blk - nb_zoh) * blockDim.x + threadIdx.x;
        const long per = (long)nint * LPI;
        const bool live = gid < (long)B * per;
        if (!live) gid = (long)B * per - LPI + (gid & (LPI - 1));
        const int b = (int)(gid / per); const int rem = (int)(gid - (long)b * per);
        const int t = (zoh ? 0 : C.Ns) + rem / LPI, g = rem % LPI;
        if (zoh) linearize_lanes<6 / LPI, 6>(C, b, t, g, live, nodes, dt, hji_Mb, qp);
        else linearize_lanes<8 / LPI, 8>(C, b, t, g, live, nodes, dt, hji_Mb, qp);
    }
}
#ifndef PG_LIN_WAVES
#define PG_LIN_WAVES 1
#endif
template <int K>
__global__ __launch_bounds__(64, PG_LIN_WAVES) void k_linearize(DevCfg C, int B, const real* __restrict__ nodes, const tdouble* __restrict__ dt, const real* __restrict__ hji_Mb, real* __restrict__ qp) {
    constexpr int G = 8 / K;
    long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long per = (long)C.N * G;
    const bool live = gid < (long)B * per;
    if (!live) gid = (long)B * per - G + (gid & (G - 1));        // keep the lane group whole for the shuffles; dead lanes store nothing
    const int b = (int)(gid / per); const int rem = (int)(gid - (long)b * per);
    linearize_lanes<K>(C, b, rem / G, rem % G, live, nodes, dt, hji_Mb, qp);
}

// ------------------------------------------------------------------------------------------------------------------
// compute_linearization_nodes! and update_QP! of a batch with cold instances as ONE launch (pg_step_dev / pg_simulate_dev, large batches; the (M, b) of a safety row
// are computed before it).
// The cold seeding is a serial recurrence over the 31 nodes of an instance (lane = instance: 64 wavefronts at B = 4096, 0.18 ms of pure latency on 64 of the
// 1024 SIMDs), and `linearize` of interval t reads nodes t and t + 1 only -- so the linearisation of the early intervals can run while the recurrence is still
// on its way down the horizon.  Blocks [0, nb_nodes) run the nodes recurrence (the body of k_nodes) and publish their progress after every node; the blocks
// behind them are the linearisation in INTERVAL-major order (block = 64 / LPI instances x LPI lanes of one interval t; t < Ns: six directions, else eight --
// the two instruction streams of k_linearize_split<LPI>) and wait until nodes t, t + 1 of their instances are published.  The same device functions on the same
// arguments as the two-launch sequence: bit-identical nodes and QP data.
//  * Forward progress: workgroups are dispatched in index order, so the nodes blocks are resident before any waiting block (nb_nodes <= 256 of 1024 SIMD slots is
//    required by the host); they never wait for anything.  A waiting wavefront sleeps between polls and gives up after ~0.1 s: it then poisons its share of the
//    QP data with NaN (k_solve reports PG_NUMERICAL for those instances) and exits, so the grid drains whatever happens.
//  * The angles k_nodes defers to k_nodes_angles (delta of every seeded node, -beta of the long ones) are finished by the linearisation lanes from the same
//    arguments (`naux`) and written into the nodes array by the lanes of the interval that starts at the node (and the last interval for node N).
//  * Visibility: release (fence + relaxed store at device scope) on the producer side, relaxed polls + acquire fence on the consumer side.  A device-scope release
//    writes the dirty lines of the XCD's L2 back -- including the QP data the linearisation wavefronts of that XCD are in the middle of writing -- so the recurrence
//    publishes after a few chosen nodes (`pub_mask`), not after every node: measured, B = 16384: 1.595 ms with 30 publications per wavefront, 1.434 with 6 (two launches:
//    1.45); fp32 at 8192: 0.572 / 0.483 (0.555); B = 4096: 0.41 / 0.40 (0.53).
// WAVES (fp32 only): 2 = held to two waves per SIMD.  The fp32 kernel sits within a register or two of 256 and falls to one wave with any small change; with two, 8192
// instances per GPU run 0.342 against 0.383 ms, 4096 -- one round of wavefronts either way, each slower next to a neighbour -- 0.258 against 0.249: the host picks by batch size.
template <bool STAGED, int LPI, int WAVES = 1> __global__ __launch_bounds__(64, WAVES) void k_nodes_linearize(DevCfg C, int B, int nb_nodes, int nz_first, unsigned long long pub_mask, const real* __restrict__ state, const real* __restrict__ control,
                        const tdouble* __restrict__ toff, const int* __restrict__ solved, const real* __restrict__ sep, const tdouble* __restrict__ ts, const tdouble* __restrict__ dt,
                        const tdouble* __restrict__ prev_ts, const real* __restrict__ prev_x, real* nodes, OrderOut F, real* naux, int* progress, const real* __restrict__ hji_Mb, real* __restrict__ qp, int* fault, int* fault_total) {
    PG_NL_MARK(0, wall_clock64());
    if ((int)blockIdx.x < nb_nodes) {
        nodes_body<STAGED, true>(C, B, (int)blockIdx.x, state, control, toff, solved, sep, ts, dt, prev_ts, prev_x, nodes, F, naux, progress, pub_mask);
        PG_NL_MARK(2, wall_clock64()); PG_NL_MARK(3, 1000ull);
        __threadfence();                                                  // (warm or mixed wavefronts publish once, here; every lane is back from the body)
        // (bit 63 of pub_mask = fault injection, option "diag_pipe_fault" of the -DPG_DIAG build: the recurrence never publishes, so that a test can watch every waiting wavefront give up)
        if (threadIdx.x == 0 && !(pub_mask >> 63)) __hip_atomic_store(progress + blockIdx.x, C.NN, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    extern __shared__ real sh_rec[];                                      // (the dynamic LDS of the launch: max(trajectory channels, 64 x 20 node-record entries))
    // a block = one interval t of 64 / LPI instances, LPI lanes each.  Order of the intervals: the first nz_first intervals of the short horizon (enough wavefronts to
    // fill the machine while the recurrence is still in the short horizon), then the ramp intervals of the long horizon, and the remaining short-horizon
    // intervals LAST -- the launch ends with its cheapest wavefronts (0.408 -> 0.399 ms; grouping several intervals of an instance in a wavefront: no difference)
    // (measured and dropped, EXPERIMENTS.md 10.4: lane pairs for the LAST ramp intervals -- the shorter chain behind the recurrence's last publication -- and for the FIRST
    // short-horizon intervals -- more wavefronts while most SIMDs are still idle: 0.337 -> 0.351 / 0.343 ms.  One lane per interval everywhere.)
    constexpr int IPB = 64 / LPI;                                         // instances per block
    const int w = (int)blockIdx.x - nb_nodes, nbt = (B + IPB - 1) / IPB;
    int t = w / nbt; const int grp = w - t * nbt;
    if (t >= nz_first) t = t < nz_first + (C.N - C.Ns) ? t - nz_first + C.Ns : t - (C.N - C.Ns);
    const int lane = (int)threadIdx.x, g = lane % LPI;
    int b = grp * IPB + lane / LPI;
    const bool live = b < B;
    if (!live) b = B - 1;
    {
        // (the nodes wavefronts of these instances: a block of 64 / LPI instances is seeded by one or more nodes blocks of NODES_IPB instances -- it waits for the slowest)
        constexpr int NFL = IPB > NODES_IPB ? IPB / NODES_IPB : 1;
        const int f0 = (grp * IPB) / NODES_IPB;
        const int* flag = progress + f0;
        const int nfl = f0 + NFL <= nb_nodes ? NFL : nb_nodes - f0;
        const int need = t + 2;                                             // nodes 0 .. t + 1
        // The wait is bounded in WALL-CLOCK time (s_memrealtime: 100 MHz whatever the shader clock does): 20 ms -- a hundred times the whole recurrence.  A wavefront that
        // gives up (the nodes blocks not resident before it: a dispatch order this launch does not control; a debugger, a time-sliced or serialised profiler run)
        // raises `fault` and leaves; every other waiting wavefront then leaves at once as well, and the host has queued, behind this launch, the ordinary
        // k_nodes_angles + k_linearize_split over the whole batch, predicated on that word: the step is late, never wrong.
        const unsigned long long t_give_up = wall_clock64() + 2000000ull;
        bool gave_up = false;
        auto seeded = [&]() { int m = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                              for (int q = 1; q < NFL; q++) if (q < nfl) { const int v = __hip_atomic_load(flag + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); m = v < m ? v : m; }
                              return m; };
        while (seeded() < need) {
            __builtin_amdgcn_s_sleep(16);
            if (__hip_atomic_load(fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 || wall_clock64() > t_give_up) { gave_up = true; break; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (gave_up) {
            if (lane == 0) { __hip_atomic_fetch_add(fault, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_fetch_add(fault_total, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
            return;
        }
    }
    PG_NL_MARK(1, wall_clock64()); PG_NL_MARK(3, (unsigned long long)t);
    const real* n0g = nodes + ((size_t)b * C.NN + t) * 10;
    const real* a0 = naux + ((size_t)b * C.NN + t) * 4;
    real* rec = sh_rec + lane * 20;
#pragma unroll
    for (int k = 0; k < 20; k++) rec[k] = n0g[k];
    {   // k_nodes_angles, for the two nodes of this interval
        const real y0 = a0[0], x0 = a0[1], t0 = a0[2], b0 = a0[3], y1 = a0[4], x1 = a0[5], t1 = a0[6], b1 = a0[7];
        if (y0 == y0) {
            rec[6] = atan2(y0, x0) - atan(t0);
            if (b0 == b0) rec[4] = -atan(b0);
            if (live && g == 0) { real* nd = nodes + ((size_t)b * C.NN + t) * 10; nd[6] = rec[6]; nd[4] = rec[4]; }
        }
        if (y1 == y1) {
            rec[16] = atan2(y1, x1) - atan(t1);
            if (t == C.N - 1) {
                if (b1 == b1) rec[14] = -atan(b1);
                if (live && g == LPI - 1) { real* nd = nodes + ((size_t)b * C.NN + t + 1) * 10; nd[6] = rec[16]; nd[4] = rec[14]; }
            }
        }
    }
    if (t < C.Ns) linearize_lanes_at<6 / LPI, 6>(C, b, t, g, live, rec, rec + 10, dt, hji_Mb, qp);
    else linearize_lanes_at<8 / LPI, 8>(C, b, t, g, live, rec, rec + 10, dt, hji_Mb, qp);
    PG_NL_MARK(2, wall_clock64());
}

// ------------------------------------------------------------------------------------------------------------------
// Closed-loop plant step of `simulate` (model_predictive_control.jl:94-95), lane = instance:
//   state   <- propagate(dynamics, state, StepControl(dt, BicycleControl2(current_control)))   (RK4, nsub sub-steps, world-frame BicycleModel
//              through the actuator limits: vehicle_dynamics.jl:111-135,293-314)
//   control <- get_next_control(mpc)     (one-step actuation delay: the state moves with the OLD control)
//   t       <- the next element of the loop's range `0:dt:trajectory.t[end]` (:87), shifted by the instance's start time: (t_start .+ clk)[idx] -- ONE rounding of
//              t_start + (idx - 1) dt with dt lifted to its rational, as Julia's range gives it (rounds 1-5 accumulated t += dt: option "time_grid_naive")
__global__ __launch_bounds__(64) void k_advance(DevCfg C, int B, tdouble dtp, real* __restrict__ state, real* __restrict__ control, const real* __restrict__ u_next,
                                                tdouble* __restrict__ t0, const tdouble* __restrict__ t_start, JlRange clk, int idx) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    real* q = state + (size_t)b * 6; real* u = control + (size_t)b * 3;
    real x[6] = {q[0], q[1], q[2], q[3], q[4], q[5]};
    const real d = u[0], Fx = u[1] + u[2];
    const int nsub = C.nsub; const real h = dtp / nsub;
    auto rhs = [&](const real* y, real* o) {
        real s, c; pg_sincos(y[2], &s, &c);
        o[0] = -y[3] * s - y[4] * c; o[1] = y[3] * c - y[4] * s; o[2] = y[5];          // psi measured from North (:127-129)
        world_body_rhs<real>(C.veh, y[3], y[4], y[5], d, Fx, o[3], o[4], o[5]);
    };
#pragma unroll 1
    for (int i = 0; i < nsub; i++) {
        real k1[6], k2[6], k3[6], k4[6], y[6];
        rhs(x, k1);
        for (int k = 0; k < 6; k++) y[k] = x[k] + k1[k] * (h * real(0.5));
        rhs(y, k2);
        for (int k = 0; k < 6; k++) y[k] = x[k] + k2[k] * (h * real(0.5));
        rhs(y, k3);
        for (int k = 0; k < 6; k++) y[k] = x[k] + k3[k] * h;
        rhs(y, k4);
        for (int k = 0; k < 6; k++) x[k] += (k1[k] + real(2.0) * k2[k] + real(2.0) * k3[k] + k4[k]) * (h / real(6.0));
    }
    for (int k = 0; k < 6; k++) q[k] = x[k];
    u[0] = u_next[(size_t)b * 3]; u[1] = u_next[(size_t)b * 3 + 1]; u[2] = u_next[(size_t)b * 3 + 2];
    t0[b] = C.time_grid_naive ? t0[b] + dtp : jl_shifted_elem(clk, t_start[b], idx);
}

// ==================================================================================================================
// Decoupled (lateral) formulation: decoupled_lat_long.jl.  The lateral QP (state (Uy, r, dpsi, e), input delta) is EMBEDDED in the
// 8-state stage structure k_solve works on: x = (0, Ux_dummy, Uy, r, dpsi, e, delta, 0) with identity dynamics, zero cost and
// never-active bounds on the three inert slots, so the same solve kernel serves both formulations (the inert slots are exactly
// decoupled from the rest, the optimum of the embedded problem restricted to the live slots IS the lateral optimum).
// Node record (10 doubles): (0, Ux parameter, Uy, r, dpsi, e, delta, Fx, 0, kappa).
// Lanes per instance in k_nodes_dec (round 6).  The seeding of the lateral formulation is the same recurrence as the coupled one's cold branch, run at EVERY step (there is no warm
// branch: decoupled_lat_long.jl:52-104) over 51 nodes at N = 50: one lane per instance took 0.31-0.33 ms -- half of a settled closed-loop step.  As in nodes_body: the long
// nodes (four-iteration solves from (V kappa, 0, 0, 0)) are coupled through ONE number, the acceleration the solve returns, and that is the commanded acceleration pulled back
// onto the friction circle unless a limit binds inside the solve.  LPN lanes run the cheap (V, s) chain LPN nodes ahead on that value, each solves ITS node, a ballot commits the
// leading run of nodes whose solve returned the value assumed (to 1e-12), and the state behind the last committed node advances with that node's own solve value, as the serial
// form does.  Option "nodes_serial": one node per pass -- the serial form exactly.
#ifndef PG_NODES_DEC_LPN
#define PG_NODES_DEC_LPN 8
#endif
constexpr int NODES_DEC_LPN = PG_NODES_DEC_LPN, NODES_DEC_IPB = 64 / NODES_DEC_LPN;
template <bool STAGED> __global__ __launch_bounds__(64) void k_nodes_dec(DevCfg C, int B, const real* __restrict__ state, const real* __restrict__ control, const tdouble* __restrict__ toff,
                                                  const real* __restrict__ sep, const tdouble* __restrict__ ts, const tdouble* __restrict__ dt, real* __restrict__ nodes, real* __restrict__ naux) {
    // As in k_nodes: the arclength and time channels are searched in ONE lockstep loop per node (traj_lookup2; the wall edges reuse its knot), and the three inverse tangents of a
    // seeded node -- they feed nothing in the chain -- are left to k_nodes_angles (lane = (instance, node)), launched behind this kernel.
    constexpr int LPN = NODES_DEC_LPN;
    extern __shared__ real sh_traj[];
    TrajView T = C.traj;
    if constexpr (STAGED) {                                 // single shared trajectory; compile-time so the searches compile to ds_read, not flat loads
        stage_trajectory(C.traj, sh_traj);
        __syncthreads();
        T.t = sh_traj; T.s = sh_traj + T.L;
    }
    const int b = blockIdx.x * NODES_DEC_IPB + (int)threadIdx.x / LPN, lp = (int)threadIdx.x % LPN;
    if (b >= B) return;
    const bool lead = lp == 0;                              // (the lane that stores what all lanes of the instance compute alike)
    if constexpr (!STAGED) T = traj_of(C, b);
    const DevVehicle& P = C.veh;
    const real* q0 = state + (size_t)b * 6; const real* u0 = control + (size_t)b * 3;
    const tdouble* TS = ts + (size_t)b * C.NN; const tdouble* DT = dt + (size_t)b * C.N;
    real* ND = nodes + (size_t)b * C.NN * 10;
    real s = sep[(size_t)b * 4]; const real e0 = sep[(size_t)b * 4 + 1];           // :65
    const real psi0 = q0[2], Ux0 = q0[3], Uy0 = q0[4], r0 = q0[5], d0 = u0[0], Fxf0 = u0[1], Fxr0 = u0[2];
    real V = hypot(Ux0, Uy0);                                                         // :67
    const real beta0 = atan2(Uy0, Ux0);
    real Fyf0, Fyr0, sb0, cb0, sd0, cd0;
    pg_sincos(beta0, &sb0, &cb0); pg_sincos(d0, &sd0, &cd0);
    {
        real sd = sd0, cd = cd0;
        real af = tan(atan2(Uy0 + P.a * r0, Ux0) - d0), ar = tan(atan2(Uy0 - P.b * r0, Ux0));
        lateral_forces<real>(P, af, ar, Fxf0, Fxr0, sd, cd, Fyf0, Fyr0);               // :71
    }
    const bool traj_mode = !(toff[b] != toff[b]);
    auto put_walls = [&](int i, int jk, real wk) __attribute__((always_inline)) {      // traj_edges_at_s(T, s): the knot and weight of the node's lookup
        real* w = C.wall_edges + ((size_t)b * C.N + i - 1) * 2;
        w[0] = T.edge_L[jk] + wk * (T.edge_L[jk + 1] - T.edge_L[jk]); w[1] = T.edge_R[jk] + wk * (T.edge_R[jk + 1] - T.edge_R[jk]);
    };
    // the measured node and the short horizon (one iteration from the measured state: not a fixed point -- serial, every lane of the instance alike, the lead lane stores)
    const bool ahead = LPN > 1 && C.Ns + 1 < C.NN;
    const int i_serial_end = ahead ? C.Ns + 1 : C.NN;
#pragma unroll 1
    for (int i = 0; i < i_serial_end; i++) {
        const real tau = (i == C.NN - 1) ? DT[i - 1] : DT[i];
        TrajS tj; real s_ref; int jk; real wk;
        traj_lookup2(T, s, TS[i], tj, s_ref, &jk, &wk);
        const real A_des = commanded_accel(C.cp, tj.A, tj.V, V, s - s_ref, tau, traj_mode);                                   // :76-77
        NodeRec r; real A;
        real* ax = naux + ((size_t)b * C.NN + i) * 4;
        r.q0 = real(0.0); r.pV = real(0.0); r.pK = tj.kappa;
        real a0 = NAN, a1 = real(0.0), a2 = real(0.0), a3 = NAN;
        if (i == 0) {
            r.q1 = Ux0; r.q2 = Uy0; r.q3 = r0; r.q4 = adiff(psi0, tj.psi); r.q5 = e0; r.u0 = d0; r.u1 = Fxf0 + Fxr0;      // :79-81
            real dUx, dUy, dr;
            world_body_rhs<real>(P, Ux0, Uy0, r0, d0, Fxf0 + Fxr0, dUx, dUy, dr);                                          // :82
            A = (dUx - r0 * Uy0) * cb0 + (dUy + r0 * Ux0) * sb0;                                                             // :83   (a0 = NaN: the measured state, nothing deferred)
        } else {
            const bool shortp = i <= C.Ns;
            Steady est = steady_state(P, V, A_des, tj.kappa, shortp ? 1 : 4, shortp ? r0 : V * tj.kappa, shortp ? beta0 : real(0.0), shortp ? sb0 : real(0.0), shortp ? cb0 : real(1.0),
                                      shortp ? d0 : real(0.0), shortp ? sd0 : real(0.0), shortp ? cd0 : real(1.0), shortp ? Fyf0 : real(0.0), true);
            r.q1 = est.Ux;
            r.q2 = shortp ? Uy0 : est.Uy; r.q3 = shortp ? r0 : est.r; r.q4 = shortp ? adiff(psi0, tj.psi) : -est.beta; r.q5 = shortp ? e0 : real(0.0);   // :85,92 (long nodes: q4 = -beta by k_nodes_angles)
            r.u0 = real(0.0); r.u1 = est.Fx; A = est.A;                                                                      // (u0 = delta by k_nodes_angles)
            a0 = est.ang_y; a1 = est.ang_x; a2 = est.ang_t; a3 = (!shortp && est.beta_is_tan) ? est.tb : NAN;
        }
        if (lead) {
            put_node(ND, i, r);
            ax[0] = a0; ax[1] = a1; ax[2] = a2; ax[3] = a3;
            if (C.walls && i >= 1) put_walls(i, jk, wk);
        }
        advance_vs(V, s, A, tau);                                                                                           // :100-101
    }
    if (ahead) {
        const real tolA = sizeof(real) == 8 ? real(1e-12) : real(4e-6);
        const int base = (int)(threadIdx.x & 63u) - lp;        // first lane of this instance in the wavefront
        int i = C.Ns + 1;
#pragma unroll 1
        while (__any(i < C.NN)) {
            // (V, s) ahead on the commanded acceleration pulled onto the friction circle: every lane of the instance runs the same cheap chain and keeps the arguments of ITS node
            real Vs = V, ss = s, Vm = V, Adm = real(0.0), kpm = real(0.0), Agm = real(0.0), wkm = real(0.0); int jkm = 0;
            real Vst[LPN], sst[LPN], tauj[LPN];
#pragma unroll
            for (int j = 0; j < LPN; j++) {
                const int ii = i + j < C.NN ? i + j : C.NN - 1;
                const real tau = (ii == C.NN - 1) ? DT[ii - 1] : DT[ii];
                TrajS tjj; real s_ref; int jk; real wk;
                traj_lookup2(T, ss, TS[ii], tjj, s_ref, &jk, &wk);
                const real A_des = commanded_accel(C.cp, tjj.A, tjj.V, Vs, ss - s_ref, tau, traj_mode);
                real At = A_des, Ar; limit_accel(P, Vs, tjj.kappa, At, Ar);
                if (j == lp) { Vm = Vs; Adm = A_des; kpm = tjj.kappa; Agm = At; jkm = jk; wkm = wk; }
                Vst[j] = Vs; sst[j] = ss; tauj[j] = tau;
                advance_vs(Vs, ss, At, tau);
            }
            // this lane's node: the four-iteration solve from (V kappa, 0, 0, 0) (:91), exactly as the serial form evaluates it
            const Steady est = steady_state(P, Vm, Adm, kpm, 4, Vm * kpm, real(0.0), real(0.0), real(1.0), real(0.0), real(0.0), real(1.0), real(0.0), true);
            const bool match = fabs(est.A - Agm) <= tolA * (real(1.0) + fabs(Agm));       // (a NaN never matches: the solve's own value goes on, as in the serial form)
            const unsigned long long mb = __ballot(match);
            int c = 1;      // nodes i .. i + c - 1 are valid: node i always, node i + j while every node before it returned the acceleration assumed
#pragma unroll
            for (int j = 1; j < LPN; j++) c += (c == j && ((mb >> (base + j - 1)) & 1ull) && i + j < C.NN && !C.nodes_serial) ? 1 : 0;
            if (i < C.NN && lp < c) {
                NodeRec rr;
                rr.q0 = real(0.0); rr.q1 = est.Ux; rr.q2 = est.Uy; rr.q3 = est.r; rr.q4 = -est.beta; rr.q5 = real(0.0);      // (q4 = -beta and u0 = delta are finished by k_nodes_angles)
                rr.u0 = real(0.0); rr.u1 = est.Fx; rr.pV = real(0.0); rr.pK = kpm;
                put_node(ND, i + lp, rr);
                real* ax = naux + ((size_t)b * C.NN + i + lp) * 4;
                ax[0] = est.ang_y; ax[1] = est.ang_x; ax[2] = est.ang_t; ax[3] = est.beta_is_tan ? est.tb : NAN;
                if (C.walls) put_walls(i + lp, jkm, wkm);
            }
            if (i < C.NN) {      // the state behind the last valid node, from the acceleration ITS solve returned (every lane of the instance alike)
                const int jl = c - 1;
                const real A_last = __shfl(est.A, base + jl);
                real Vs0 = Vst[0], ss0 = sst[0], tl = tauj[0];
#pragma unroll
                for (int j = 1; j < LPN; j++) if (j == jl) { Vs0 = Vst[j]; ss0 = sst[j]; tl = tauj[j]; }
                V = Vs0; s = ss0; advance_vs(V, s, A_last, tl);
                i += c;
            }
        }
    }
}

// 4x4 helpers for the exact discretisation
struct M4d { real a[16]; };
PG_DEV M4d m4mul(const M4d& x, const M4d& y) { M4d r;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) { real s = real(0.0);
#pragma unroll
            for (int k = 0; k < 4; k++) s += x.a[4 * i + k] * y.a[4 * k + j];
            r.a[4 * i + j] = s; }
    return r; }

// update_QP! of the lateral formulation (decoupled_lat_long.jl:228-273), lane = (instance, interval):
// continuous Jacobians by forward mode (8 tangent directions: Uy, r, dpsi, e, delta, Fx, Ux, kappa), exact ZOH / FOH discretisation
// (Ad = exp(A dt), G0 = int exp(A s) ds, G1 = (1/dt) int exp(A (dt - s)) s ds by Taylor series + scaling and squaring), envelope and bounds;
// the result is written in the embedded coupled layout (QP block + the packed per-stage block k_solve streams).
// embed = 0 (round 5; the steps of a handle whose solver is k_solve_lat): only the packed stage records and the fixed first node are written -- 92 MB instead of 283 MB per
// 4096-instance launch of a kernel that is bound by its writes; pg_get_qp re-runs the kernel with embed = 1 when somebody asks for the embedded block.
__global__ __launch_bounds__(128) void k_qp_dec(DevCfg C, int B, const real* __restrict__ nodes, const tdouble* __restrict__ dt, real* __restrict__ qp, int embed) {
    long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long)B * C.N) return;
    int b = (int)(gid / C.N), t = (int)(gid - (long)b * C.N);
    const bool ramp = t >= C.Ns;
    const real* n0 = nodes + ((size_t)b * C.NN + t) * 10; const real* n1 = n0 + 10;
    const real q[4] = {n0[2], n0[3], n0[4], n0[5]};
    const real w0[4] = {n0[6], n0[7], n0[1], n0[9]}, wf[4] = {n1[6], n1[7], n1[1], n1[9]};      // (delta, Fx, Ux, kappa); theta = phi = 0 carry no derivative
    const real T = dt[(size_t)b * C.N + t];
    M4d A; real Bc[4][4], cc[4];
    {
        D2 f[4];
        D2 x[4] = {D2(q[0], real(1.0), real(0.0)), D2(q[1], real(0.0), real(1.0)), D2(q[2]), D2(q[3])};
        lateral_rhs<D2>(C.veh, x, D2(w0[0]), D2(w0[1]), D2(w0[2]), D2(w0[3]), f);
        for (int i = 0; i < 4; i++) { A.a[4 * i] = f[i].a; A.a[4 * i + 1] = f[i].b; cc[i] = f[i].v; }
        D2 y[4] = {D2(q[0]), D2(q[1]), D2(q[2], real(1.0), real(0.0)), D2(q[3], real(0.0), real(1.0))};
        lateral_rhs<D2>(C.veh, y, D2(w0[0]), D2(w0[1]), D2(w0[2]), D2(w0[3]), f);
        for (int i = 0; i < 4; i++) { A.a[4 * i + 2] = f[i].a; A.a[4 * i + 3] = f[i].b; }
        D2 z[4] = {D2(q[0]), D2(q[1]), D2(q[2]), D2(q[3])};
        lateral_rhs<D2>(C.veh, z, D2(w0[0], real(1.0), real(0.0)), D2(w0[1], real(0.0), real(1.0)), D2(w0[2]), D2(w0[3]), f);
        for (int i = 0; i < 4; i++) { Bc[i][0] = f[i].a; Bc[i][1] = f[i].b; }
        lateral_rhs<D2>(C.veh, z, D2(w0[0]), D2(w0[1]), D2(w0[2], real(1.0), real(0.0)), D2(w0[3], real(0.0), real(1.0)), f);
        for (int i = 0; i < 4; i++) { Bc[i][2] = f[i].a; Bc[i][3] = f[i].b; }
        for (int i = 0; i < 4; i++) {
            real ci = cc[i];
            for (int j = 0; j < 4; j++) ci -= A.a[4 * i + j] * q[j] + Bc[i][j] * w0[j];
            cc[i] = ci;                                                                    // c = f - A x - B w
        }
    }
    // exp(A T) and its integrals
    real nrm = real(0.0);
    for (int i = 0; i < 4; i++) { real sr = real(0.0); for (int j = 0; j < 4; j++) sr += fabs(A.a[4 * i + j]); nrm = fmax(nrm, sr); }
    int sq = 0; real h = T;
    while (nrm * h > real(0.25) && sq < 40) { h *= real(0.5); sq++; }
    M4d Ah, term, Ad, G0, G2;
    for (int i = 0; i < 16; i++) { Ah.a[i] = A.a[i] * h; real e = (i % 5 == 0) ? real(1.0) : real(0.0); term.a[i] = e; Ad.a[i] = e; G0.a[i] = e * h; G2.a[i] = e * h * h * real(0.5); }
#pragma unroll 1
    for (int k = 1; k <= 16; k++) {
        term = m4mul(term, Ah);
        const real ik = real(1.0) / k, c0 = h / (k + 1), c2 = h * h / ((k + real(1.0)) * (k + real(2.0)));
        for (int i = 0; i < 16; i++) { term.a[i] *= ik; Ad.a[i] += term.a[i]; G0.a[i] += term.a[i] * c0; G2.a[i] += term.a[i] * c2; }
    }
#pragma unroll 1
    for (int i = 0; i < sq; i++) {
        M4d AG2 = m4mul(Ad, G2), AG0 = m4mul(Ad, G0), AA = m4mul(Ad, Ad);
        for (int j = 0; j < 16; j++) { G2.a[j] = G2.a[j] + h * G0.a[j] + AG2.a[j]; G0.a[j] += AG0.a[j]; Ad.a[j] = AA.a[j]; }
        h *= real(2.0);
    }
    real b0[4], bf[4], cd[4];
    const real iT = real(1.0) / T;
    for (int i = 0; i < 4; i++) {
        real s0 = real(0.0), sf = real(0.0), sc = real(0.0);
        for (int k = 0; k < 4; k++) {
            real g0 = G0.a[4 * i + k], g1 = ramp ? G2.a[4 * i + k] * iT : real(0.0);
            s0 += (g0 - g1) * Bc[k][0]; sf += g1 * Bc[k][0];
            real fold = cc[k] * g0;
            for (int j = 1; j < 4; j++) fold += (g0 - g1) * Bc[k][j] * w0[j] + g1 * Bc[k][j] * (ramp ? wf[j] : real(0.0));
            sc += fold;
        }
        b0[i] = s0; bf[i] = sf; cd[i] = sc;
    }
    // envelope and bounds (:262-272): Ux from the NEXT node's parameter, Fx from its seeded control; nothing is normalised here
    real Uxt = n1[1], Fx = n1[7];
    real Fxf = Fx > real(0.0) ? Fx * C.veh.fwd_frac : Fx * C.veh.fwb_frac, Fxr = Fx > real(0.0) ? Fx * C.veh.rwd_frac : Fx * C.veh.rwb_frac;
    Envelope e = stable_limits(C.veh, Uxt, Fxf, Fxr);
    const real dmin_t = jmax(e.dmin, -C.veh.delta_max), dmax_t = jmin(e.dmax, C.veh.delta_max), ddmin_t = -C.cp.deltadot_max * T, ddmax_t = C.cp.deltadot_max * T;
    // ---- embedded coupled layout ----
    QpOff o = qp_offsets(C.N);
    real* Q = qp + (size_t)b * C.qp_len;
    if (embed) {
        real* A6 = Q + o.A + 36 * t; real* B06 = Q + o.B0 + 12 * t; real* Bf6 = Q + o.Bf + 12 * t; real* c6 = Q + o.c + 6 * t;
        for (int i = 0; i < 36; i++) A6[i] = real(0.0);
        for (int i = 0; i < 12; i++) { B06[i] = real(0.0); Bf6[i] = real(0.0); }
        A6[0] = real(1.0); A6[7] = real(1.0); c6[0] = real(0.0); c6[1] = real(0.0);
        for (int i = 0; i < 4; i++) {
            for (int j = 0; j < 4; j++) A6[6 * (2 + i) + 2 + j] = Ad.a[4 * i + j];
            B06[2 * (2 + i)] = b0[i]; Bf6[2 * (2 + i)] = bf[i]; c6[2 + i] = cd[i];
        }
        for (int i = 0; i < 4; i++) { Q[o.H + 8 * t + 2 * i] = e.H[i][0]; Q[o.H + 8 * t + 2 * i + 1] = e.H[i][1]; Q[o.G + 4 * t + i] = e.G[i]; }
        Q[o.dmin + t] = dmin_t; Q[o.dmax + t] = dmax_t;
        Q[o.fxmax + t] = real(1.0);                                           // inert Fx slot: 0 <= 1 is never active
        Q[o.ddmin + t] = ddmin_t; Q[o.ddmax + t] = ddmax_t;
        Q[o.dt + t] = T;
    }
    if (C.lat_pack) {      // the same numbers once more, packed for k_solve_lat
        real* Lp = C.lat_pack + ((size_t)b * C.N + t) * LATP;
        for (int i = 0; i < 4; i++) {
            for (int j = 0; j < 4; j++) Lp[8 * i + j] = Ad.a[4 * i + j];
            Lp[8 * i + 4] = b0[i] + bf[i]; Lp[8 * i + 5] = bf[i]; Lp[8 * i + 6] = cd[i]; Lp[8 * i + 7] = real(0.0);
            Lp[32 + 2 * i] = e.H[i][0]; Lp[33 + 2 * i] = e.H[i][1]; Lp[40 + i] = e.G[i];
        }
        Lp[44] = dmax_t; Lp[45] = dmin_t; Lp[46] = ddmax_t; Lp[47] = ddmin_t; Lp[48] = T;
        for (int i = 49; i < LATP; i++) Lp[i] = real(0.0);
    }
    if (t == 0) {
        Q[o.qcurr] = real(0.0); Q[o.qcurr + 1] = C.ux_dummy; for (int k = 0; k < 4; k++) Q[o.qcurr + 2 + k] = q[k];
        Q[o.ucurr] = n0[6]; Q[o.ucurr + 1] = real(0.0); Q[o.M] = real(0.0); Q[o.M + 1] = real(0.0); Q[o.b] = real(1.0);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// HJI grid on device: node record = 8 floats (V, gradV[0..6]) -> one 32 B aligned read per corner, 64 B per dim-1 pair.
// Lookup layout: CELL records.  A cell of the leading `cdims` dimensions stores its 2^cdims corner nodes contiguously, ordered by the
// corner bits (b1 + 2 b2 + ...), 32 B per node.  cdims = 5: 1 KiB records, a 16-lane group reads one record as a single coalesced 1 KiB
// access (4 records per lookup); cdims = 3: 256 B records, one per lane (16 per lookup).  Cells overlap, i.e. every node is stored up
// to 2^cdims times (10 GB / 2.6 GB for the 10 M-node grid): on a 288 GB part HBM capacity is not the constraint, DRAM-page and line
// efficiency of a random gather is.  pg_set_hji_grid picks the largest cdims whose table fits the memory budget.
struct HjiView { int dims[7]; int koff[7]; long stride[7]; long cstride[7]; int cdims; const float* knots; const float* nodes; const float* cells; };

// builds the cell records from the compact node records: thread = (cell, corner)
__global__ __launch_bounds__(256) void k_hji_build_cells(HjiView Hv, long ncell, float* __restrict__ cells) {
    long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int cd = Hv.cdims;
    if (gid >= (ncell << cd)) return;
    long cell = gid >> cd; int corner = (int)(gid & ((1 << cd) - 1));
    long rem = cell, node = 0;
#pragma unroll
    for (int d = 0; d < 7; d++) {
        int ext = d < cd ? Hv.dims[d] - 1 : Hv.dims[d];
        int i = (int)(rem % ext); rem /= ext;
        if (d < cd) i += (corner >> d) & 1;
        node += (long)i * Hv.stride[d];
    }
    const float4* src = reinterpret_cast<const float4*>(Hv.nodes + node * 8);
    float4* dst = reinterpret_cast<float4*>(cells + gid * 8);
    dst[0] = src[0]; dst[1] = src[1];
}

// HJIRelativeState(us, them): HJI_computation.jl:20-24 (cpsi = sin(-psi), spsi = cos(-psi): names swapped in the reference)
__global__ __launch_bounds__(256) void k_hji_relstate(int B, const real* __restrict__ state, const real* __restrict__ other, real* __restrict__ x7) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const real* us = state + (size_t)b * 6; const real* th = other + (size_t)b * 4; real* x = x7 + (size_t)b * 7;
    real s, c; pg_sincos(-us[2], &s, &c);
    real cpsi = s, spsi = c, dE = th[0] - us[0], dN = th[1] - us[1];
    x[0] = cpsi * dE + spsi * dN; x[1] = -spsi * dE + cpsi * dN; x[2] = adiff(th[2], us[2]);
    x[3] = us[3]; x[4] = us[4]; x[5] = th[3]; x[6] = us[5];
}

// cache[x]: HJI_computation.jl:66-72.  SIXTEEN lanes per lookup (four lookups per wave): lane g of a group owns the corner bits of
// dims 4..7 and gathers ONE cell record (the 8 corners of dims 1..3 = 256 contiguous, aligned bytes: 16 x dwordx4).  Weights and sums in fp64 (Float32 grid x Float64 query); the 8 channels are reduced over
// the 16-lane row with DPP butterflies (no LDS crossbar).  out8[b] = (V, gradV[0..6]); out of bounds => (Inf, 0) (:70).
// bit-level lane moves: one version per arithmetic type
template <int CTRL> PG_DEV int dpp_mov(int x) { return __builtin_amdgcn_mov_dpp(x, CTRL, 0xF, 0xF, true); }
template <int CTRL> PG_DEV real dpp_move(real v) {
#ifdef PG_F32
    return __int_as_float(dpp_mov<CTRL>(__float_as_int(v)));
#else
    return __hiloint2double(dpp_mov<CTRL>(__double2hiint(v)), dpp_mov<CTRL>(__double2loint(v)));
#endif
}
PG_DEV real dpp_add(real v, int ctrl_is /*0: xor1, 1: xor2, 2: half mirror, 3: row mirror, 4: rotate 4, 5: rotate 8*/) {
    real p;
    if (ctrl_is == 0) p = dpp_move<0xB1>(v);          // quad_perm [1,0,3,2]
    else if (ctrl_is == 1) p = dpp_move<0x4E>(v);     // quad_perm [2,3,0,1]
    else if (ctrl_is == 2) p = dpp_move<0x141>(v);    // row_half_mirror
    else if (ctrl_is == 3) p = dpp_move<0x140>(v);    // row_mirror
    else if (ctrl_is == 4) p = dpp_move<0x124>(v);    // row_ror:4
    else p = dpp_move<0x128>(v);                      // row_ror:8
    return v + p;
}
template <int CD>
__global__ __launch_bounds__(256) void k_hji_lookup(HjiView Hv, int B, const real* __restrict__ x7, real* __restrict__ out8) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int g = threadIdx.x & 15;
    long look = gid >> 4;
    const bool live = look < B;
    if (!live) look = B - 1;                       // keep whole rows active for the DPP reduction; results of dead groups are not stored
    const real* x = x7 + (size_t)look * 7;
    // lane d < 7 of the group searches dimension d (searchsortedlast, clamp to [1, n-1], in-bounds test :67; knot vectors are tiny and L1-resident) and
    // the seven (index, weight, in-bounds) triples are shared over the 16-lane row: one 4-probe search per lane instead of seven in a row before the gather
    int idx[7]; real w[7]; bool inb = true;
    {
        const int d = g < 7 ? g : 6;
        const float* k = Hv.knots + Hv.koff[d]; const int n = Hv.dims[d]; const real xv = x[d];
        const int in_m = ((real)k[0] <= xv) && (xv <= (real)k[n - 1]);
        int lo = 0, hi = n;
        while (lo < hi) { int mid = (lo + hi) >> 1; if ((real)k[mid] <= xv) lo = mid + 1; else hi = mid; }
        const int i = lo < 1 ? 1 : (lo > n - 1 ? n - 1 : lo);
        const real k0 = k[i - 1], k1 = k[i];
        const real w_m = (xv - k0) / (k1 - k0);
#pragma unroll
        for (int e = 0; e < 7; e++) { idx[e] = __shfl(i - 1, e, 16); w[e] = __shfl(w_m, e, 16); inb = inb && (__shfl(in_m, e, 16) != 0); }
    }
    real acc[8];
#pragma unroll
    for (int c = 0; c < 8; c++) acc[c] = real(0.0);
    if (inb) {
        // The 128 corners of a lookup are 256 float4s -- corner bits (b1..b7), two float4s (channels 0..3, 4..7) per corner -- held as ONE 4 KiB record (cdims = 7),
        // FOUR 1 KiB records (cdims = 5: corner bits b6, b7 pick the record) or SIXTEEN 256 B records (cdims = 3: b4..b7 pick it), each ordered by its own corner bits.
        // In every layout lane g reads float4 number 16 j + g of the lookup (j = 0..15): a load instruction of the 16-lane group covers 256 CONTIGUOUS bytes of one
        // record, and the lane always sees the same channel half (g & 1) of corners b1..b3 = g >> 1, b4..b7 = j.
        // (Round 4: the two smaller layouts used to give every lane a record (cdims = 3) or a 64 B chunk of each record (cdims = 5) of its own -- 64 / 32 different cache
        // lines per load instruction.  They ran at 0.47 / 0.53 of the HBM peak; tools/probes/random_read_probe.hip shows that 256 B random records, read 256 B per
        // instruction and group, stream at the same 0.69-0.73 as 4 KiB ones: the layouts were bound by their address pattern, not by the memory.)
        constexpr int ROWS = 1 << (CD - 3);           // float4 rows of sixteen per record: 16, 4, 1
        long cell = 0;
#pragma unroll
        for (int d = 0; d < 7; d++) cell += (long)idx[d] * Hv.cstride[d];
        float4 r[16];
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const int c = j / ROWS, jr = j % ROWS;        // record of the lookup (corner bits of dims CD+1..7), row inside it
            long rec = cell;
#pragma unroll
            for (int d = CD; d < 7; d++) rec += (long)((c >> (d - CD)) & 1) * Hv.cstride[d];
            r[j] = (reinterpret_cast<const float4*>(Hv.cells + rec * (8 << CD)) + 16 * jr + g)[0];
        }
        const int lb = g >> 1;
        const real wlo = ((lb & 1) ? w[0] : (real(1.0) - w[0])) * ((lb & 2) ? w[1] : (real(1.0) - w[1])) * ((lb & 4) ? w[2] : (real(1.0) - w[2]));
        // tensor-product weights of dims 4..7 built as a tree (30 products instead of 80)
        real wt[16];
        wt[0] = wlo * (real(1.0) - w[3]); wt[1] = wlo * w[3];
#pragma unroll
        for (int j = 3; j >= 0; j--) { wt[j] = wt[j & 1] * ((j & 2) ? w[4] : (real(1.0) - w[4])); }
#pragma unroll
        for (int j = 7; j >= 0; j--) { wt[j] = wt[j & 3] * ((j & 4) ? w[5] : (real(1.0) - w[5])); }
#pragma unroll
        for (int j = 15; j >= 0; j--) { wt[j] = wt[j & 7] * ((j & 8) ? w[6] : (real(1.0) - w[6])); }
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const real wj = wt[j];
            acc[0] += wj * (real)r[j].x; acc[1] += wj * (real)r[j].y; acc[2] += wj * (real)r[j].z; acc[3] += wj * (real)r[j].w;
        }
    }
    // sum over the 8 lanes with the same channel half (same lane parity): xor 2 inside the quad, then rotations by 4 and 8 across quads
#pragma unroll
    for (int c = 0; c < 4; c++) { real v = acc[c]; v = dpp_add(v, 1); v = dpp_add(v, 4); v = dpp_add(v, 5); acc[c] = v; }
    if (live && g < 8) {      // lanes 0,2,4,6 hold channels 0..3, lanes 1,3,5,7 channels 4..7: lane g stores channel 4 (g & 1) + (g >> 1)
        const int c = g >> 1;  // after the butterflies every even lane has the same sums, every odd lane too
        real o = (c == 0) ? acc[0] : (c == 1 ? acc[1] : (c == 2 ? acc[2] : acc[3]));
        const int ch = 4 * (g & 1) + c;
        if (!inb) o = (ch == 0) ? INFINITY : real(0.0);
        out8[(size_t)look * 8 + ch] = o;
    }
}

// 2-D value slices for the RViz consumers (rviz.jl:23-40 update_HJI_values_marker!, :60-69 update_HJI_contour_marker!): both evaluate
// cache[HJIRelativeState(x, y, q[3], ..., q[7])].V at every knot pair (x, y) of grid dimensions 1 and 2.  k_hji_slice_queries writes those relative states
// (thread = (instance, i, j)), k_hji_lookup evaluates them (the same gather kernel as the safety row), k_hji_slice_post turns the values into the marker
// colours (value_to_RGB, rviz.jl:41-44) and into the zero-level crossings on the grid edges -- the vertex set of Contour.jl's contour(X, Y, V, 0):
// an edge carries a vertex iff exactly one of its ends is above the level (z > 0), at the linearly interpolated position.
__global__ __launch_bounds__(256) void k_hji_slice_queries(HjiView Hv, int B, const real* __restrict__ q7, real* __restrict__ x7) {
    const int n1 = Hv.dims[0], n2 = Hv.dims[1];
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long)B * n1 * n2) return;
    const int b = (int)(gid / (n1 * n2)); const int r = (int)(gid - (long)b * n1 * n2); const int i = r / n2, j = r - i * n2;
    const real* q = q7 + (size_t)b * 7; real* x = x7 + (size_t)gid * 7;
    x[0] = (real)Hv.knots[Hv.koff[0] + i]; x[1] = (real)Hv.knots[Hv.koff[1] + j];
#pragma unroll
    for (int d = 2; d < 7; d++) x[d] = q[d];
}
__global__ __launch_bounds__(256) void k_hji_slice_post(HjiView Hv, int B, const real* __restrict__ vg8, real* __restrict__ V_out, real* __restrict__ rgb, real* __restrict__ cross_x,
                                                        real* __restrict__ cross_y) {
    const int n1 = Hv.dims[0], n2 = Hv.dims[1];
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long)B * n1 * n2) return;
    const int b = (int)(gid / (n1 * n2)); const int r = (int)(gid - (long)b * n1 * n2); const int i = r / n2, j = r - i * n2;
    const real z = vg8[(size_t)gid * 8];
    V_out[gid] = z;
    if (rgb) {                                   // value_to_RGB(V, V_lo = -3, V_hi = 20, C_lo = (1, .5, 0), C_hi = (0, .5, 1))
        real xx = z < real(0.0) ? real(0.5) * (real(-3.0) - z) / real(-3.0) : real(0.5) + real(0.5) * z / real(20.0);
        xx = xx < real(0.0) ? real(0.0) : (xx > real(1.0) ? real(1.0) : xx);
        rgb[gid * 3] = real(1.0) - xx; rgb[gid * 3 + 1] = real(0.5); rgb[gid * 3 + 2] = xx;
    }
    const real x0 = (real)Hv.knots[Hv.koff[0] + i], y0 = (real)Hv.knots[Hv.koff[1] + j];
    if (cross_x && i + 1 < n1) {
        const real z1 = vg8[(size_t)(gid + n2) * 8], x1 = (real)Hv.knots[Hv.koff[0] + i + 1];
        cross_x[((size_t)b * (n1 - 1) + i) * n2 + j] = ((z > real(0.0)) != (z1 > real(0.0))) ? x0 + (real(0.0) - z) / (z1 - z) * (x1 - x0) : NAN;
    }
    if (cross_y && j + 1 < n2) {
        const real z1 = vg8[(size_t)(gid + 1) * 8], y1 = (real)Hv.knots[Hv.koff[1] + j + 1];
        cross_y[((size_t)b * n1 + i) * (n2 - 1) + j] = ((z > real(0.0)) != (z1 > real(0.0))) ? y0 + (real(0.0) - z) / (z1 - z) * (y1 - y0) : NAN;
    }
}

// optimal_disturbance (dMode=:min) HJI_computation.jl:90-131 + compute_reachability_constraint :160-170, lane = instance
__global__ __launch_bounds__(64) void k_hji_constraint(DevCfg C, int B, const real* __restrict__ x7, const real* __restrict__ vg8, const real* __restrict__ control,
                                 real* __restrict__ Mb) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const DevVehicle& P = C.veh;
    const real* x = x7 + (size_t)b * 7; const real* vg = vg8 + (size_t)b * 8; const real* g = vg + 1;
    real* o = Mb + (size_t)b * 4;
    real Vv = vg[0];
    o[3] = Vv;
    if (Vv > C.hji_eps) { o[0] = real(0.0); o[1] = real(0.0); o[2] = real(1.0); return; }           // :163-164
    real uH0, uH1;
    {
        real Ax_max = P.Fx_max / P.m, Pmx = P.Px_max / P.m, maxA = real(0.9) * P.mu * P.G;
        real Vh = x[5], lam_Ax = g[5], lam_Ay = g[2] / Vh;
        real nrm = (lam_Ax != lam_Ax || lam_Ay != lam_Ay) ? NAN : hypot(lam_Ax, lam_Ay);
        if (nrm < real(1e-3)) { uH0 = real(0.0); uH1 = real(0.0); }
        else {
            real desAx = -lam_Ax * maxA / nrm, desAy = -lam_Ay * maxA / nrm;
            real maxAx = jmin(Ax_max, Pmx / Vh), maxAy = P.kappa_max * Vh * Vh;
            if (desAx > maxAx) {
                if (fabs(desAy) < maxAy) maxAy = jmin(maxAy, sqrt(maxA * maxA - maxAx * maxAx));
                uH0 = copysign(maxAy, desAy) / Vh; uH1 = maxAx;
            } else if (fabs(desAy) > maxAy) {
                if (desAx > real(0.0)) { maxAx = jmin(sqrt(maxA * maxA - maxAy * maxAy), maxAx); uH0 = copysign(maxAy, desAy) / Vh; uH1 = maxAx; }
                else { uH0 = copysign(maxAy, desAy) / Vh; uH1 = -sqrt(maxA * maxA - maxAy * maxAy); }
            } else { uH0 = desAy / Vh; uH1 = maxAx; }
        }
    }
    const real* u = control + (size_t)b * 3;
    real uR0 = u[0], uR1 = u[1] + u[2];
    D2 dUx, dUy, dr;
    world_body_rhs<D2>(P, x[3], x[4], x[6], D2(uR0, real(1.0), real(0.0)), D2(uR1, real(0.0), real(1.0)), dUx, dUy, dr);   // relative_dynamics :77
    real s, c; pg_sincos(x[2], &s, &c);
    real f0 = x[5] * c - x[3] + x[1] * x[6], f1 = x[5] * s - x[4] - x[0] * x[6], f2 = uH0 - x[6];
    D2 Hm = g[3] * dUx + g[4] * dUy + g[6] * dr + (g[0] * f0 + g[1] * f1 + g[2] * f2 + g[5] * uH1);
    real M0 = Hm.a, M1 = Hm.b;
    o[0] = M0 * C.un0; o[1] = M1 * C.un1;                                          // coupled_lat_long.jl:345
    o[2] = Hm.v - (M0 * uR0 + M1 * uR1);                                           // :168
}

// optimal_control (uMode=:max, N=50) HJI_computation.jl:133-158 and the control selection of the ROS loop (ros_integration.jl:114-124), lane = instance.
// u2 [B][2] = (delta_opt, Fx_opt) whenever the relative state is inside the grid; u_next [B][3] = the policy's BicycleControl when it takes over
// (traj mode, use_policy, V <= eps), else the MPC control u_mpc; source: 0 MPC, 1 HJI policy, 2 V <= eps but the policy is switched off.
__global__ __launch_bounds__(64) void k_hji_policy(DevCfg C, int B, int use_policy, const real* __restrict__ x7, const real* __restrict__ vg8, const tdouble* __restrict__ toff,
                                                   const real* __restrict__ u_mpc, real* __restrict__ u2, real* __restrict__ u_next, int* __restrict__ source) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const DevVehicle& P = C.veh;
    const real* x = x7 + (size_t)b * 7; const real* vg = vg8 + (size_t)b * 8; const real* g = vg + 1;
    const real Ux = x[3], Uy = x[4], r = x[6];
    const real A = g[3] / P.m, Bc = g[4] / P.m + P.a * g[6] / P.Izz, Cc = g[4] / P.m - P.b * g[6] / P.Izz;      // :140-142
    const real d_opt = Bc >= real(0.0) ? P.delta_max : -P.delta_max;                                                   // :143
    real sd, cd; pg_sincos(d_opt, &sd, &cd);
    const real tf = (Uy + P.a * r) / Ux, td = sd / cd;
    const real taf = (tf - td) / (real(1.0) + tf * td), tar = (Uy - P.b * r) / Ux;       // slip-angle tangents (vehicle_dynamics.jl:84-85), Ux > 0
    real V_opt = -INFINITY, Fx_opt = real(0.0);
#pragma unroll 1
    for (int n = 0; n < 50; n++) {
        real frac = (real)n / real(49.0);
        real Fx = frac * P.Fx_max + (real(1.0) - frac) * P.Fx_min;
        real Fxf = Fx > real(0.0) ? Fx * P.fwd_frac : Fx * P.fwb_frac, Fxr = Fx > real(0.0) ? Fx * P.rwd_frac : Fx * P.rwb_frac;       // longitudinal_tire_forces, no limits (:148)
        real Fyf, Fyr;
        lateral_forces<real>(P, taf, tar, Fxf, Fxr, sd, cd, Fyf, Fyr);
        real V = A * Fx + Bc * Fyf + Cc * Fyr;
        if (V > V_opt) { Fx_opt = Fx; V_opt = V; }
    }
    u2[(size_t)b * 2] = d_opt; u2[(size_t)b * 2 + 1] = Fx_opt;
    const bool traj_mode = toff[b] == toff[b];
    const bool unsafe = traj_mode && vg[0] <= C.hji_eps;
    int src = unsafe ? (use_policy ? 1 : 2) : 0;
    real o0 = u_mpc[(size_t)b * 3], o1 = u_mpc[(size_t)b * 3 + 1], o2 = u_mpc[(size_t)b * 3 + 2];
    if (src == 1) {                  // BicycleControl(longitudinal_params, BicycleControl2(delta_opt, Fx_opt))  (vehicle_dynamics.jl:284)
        o0 = d_opt; o1 = Fx_opt > real(0.0) ? Fx_opt * P.fwd_frac : Fx_opt * P.fwb_frac; o2 = Fx_opt > real(0.0) ? Fx_opt * P.rwd_frac : Fx_opt * P.rwb_frac;
    }
    u_next[(size_t)b * 3] = o0; u_next[(size_t)b * 3 + 1] = o1; u_next[(size_t)b * 3 + 2] = o2;
    source[b] = src;
}

// ------------------------------------------------------------------------------------------------------------------
// QP solve: one wavefront per instance.  See tools/ipm_prototype.py for the algorithm statement and DESIGN.md for the derivation.
// State x_k = (q_k, u_k) in R^8, input v_k = u_{k+1} - u_k; 16 inequality rows per transition k (node k+1):
//   0: Ux >= V_min   1: Ux <= V_max   2: Fx >= Fx_min   3: delta <= dmax   4: delta >= dmin   5: Fx <= fxmax
//   6..9: H_i [Uy;r] - sigma_{i/2} <= G_i     10: sigma1 >= 0   11: sigma2 >= 0    12: d_delta <= ddmax   13: d_delta >= ddmin
//   14: M u + b + sigma_HJI >= 0              15: sigma_HJI >= 0        (14,15 only for nodes 1 .. min(N_HJI,Ns)-1)
struct SolveOut { real* sol_x; real* sol_sigma; real* u_out; int* status; int* iters; uint16_t* active; real* mu; int* solved; int* polish; real* lam;
                  const int* order_in; int* wfail;         // wfail: k_solve_lat's back-off word per instance (nullptr: none)
                  int* todo; int* n_todo;                  // rounds-only k_solve: instances left for the full kernel, and how many
                  const int* list; const int* n_list;      // full k_solve in list mode: the instances to solve (interior point at once), and how many
                  const int* mode;                         // split launch of k_solve: the PREVIOUS launch's count of instances that needed the interior point (a stream-ordered device word).
                                                           // Non-zero: the rounds-only kernel returns at once and the full kernel takes the whole batch in its launch order instead of the list
                  real* u_out2;                            // pg_step_dev: the caller's control array, written next to u_out (saves the device-to-device copy behind the launch); may be nullptr
                  int* n_whole;                            // split launch: counts the launches in which the full kernel took the WHOLE batch (`mode` non-zero) -- read-only option "stat_whole_batch_solves"; may be nullptr
                  // k_solve_lat's straggler hand-over (round 6; pg_solve_lat.hip): 0 off, 1 = this launch hands its unfinished instances over (to `todo`), 2 = this launch resumes them (`list`)
                  int hand_mode, hand_cap, hand_target, hand_min; int* hand_done; real* hand_r; int* hand_i; int hand_work, hand_w0;
                  int list_lo, list_hi; };                 // k_solve_lat in list mode: this launch serves the list only when list_lo <= its length (<= list_hi, when that is set): the host queues one launch per
                                                           // arrangement (one instance per wavefront / four) behind the warm attempts and the DEVICE word decides which of them runs

#define NROW 16
#define PG_POLISH_ROUNDS 6      // active-set rounds of the polish before it gives up
#ifndef PG_PROGRESS_ROWS
#define PG_PROGRESS_ROWS 4
#endif


struct StageRows {
    real t[NROW], lam[NROW], corr[NROW];
};

PG_DEV real wave_min(real v) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) { real o = __shfl_xor(v, s); v = o < v ? o : v; }
    return v;
}
PG_DEV real wave_max(real v) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) { real o = __shfl_xor(v, s); v = o > v ? o : v; }
    return v;
}
PG_DEV real wave_sum(real v) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) v += __shfl_xor(v, s);
    return v;
}

// inclusive prefix sum over the lanes of a wave
PG_DEV real wave_prefix_sum(real v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const real o = __shfl_up(v, d); v += lane >= d ? o : real(0.0); }
    return v;
}

// broadcast of lane `src` (compile-time constant) to the whole wave through SGPRs: no LDS, no barrier
PG_DEV real rl(real v, int src) {
#ifdef PG_F32
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
#else
    int lo = __builtin_amdgcn_readlane(__double2loint(v), src), hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
#endif
}
// waves per SIMD the solve kernel is compiled for: the fp32 iterate needs half the registers and half the LDS, so two waves share a SIMD
#ifdef PG_F32
#ifndef PG_F32_WAVES
#define PG_F32_WAVES 2          // (-DPG_F32_WAVES=1 builds the fp32 kernel for one wave per SIMD: the experiment that isolates the occupancy effect)
#endif
// (round 4: two for the rounds-only instantiation only.  The full kernel -- interior point included -- needs 700 B of scratch per lane inside 256 registers; at one wave per
//  SIMD it has none, and the launches that use it are the ones whose duration is an interior-point instance: config 3, 0.680 -> 0.639 ms)
#define PG_SOLVE_WAVES(RING, IPM) ((IPM) ? 1 : PG_F32_WAVES)
#else
#define PG_SOLVE_WAVES(RING, IPM) 1     // (the ring variant used to ask for two waves per SIMD: its LDS footprint, 30 KB at N = 50, allows five waves per CU, i.e. one per
                                        // SIMD anyway, and the 256-VGPR budget cost it 452 spilled registers: 8.3 -> 6.9 ms on the N = 50 lateral batch.  A deeper register
                                        // prefetch of the stage blocks, 2 / 4 / 6 loads in flight instead of one, was measured on top: 7.4 / 7.6 / 7.8 ms -- not load-bound)
#endif

// Ordering point between an LDS write and the LDS reads of OTHER lanes of the same wavefront.  k_solve's workgroup is one wave and the LDS pipeline
// executes the DS instructions of a wave in order, so no s_barrier and no s_waitcnt lgkmcnt(0) are needed: the compiler only has to keep the order
// (wavefront-scope fences + a scheduling barrier).  The write and the dependent reads then queue back to back instead of one round trip apart.
PG_DEV void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

#ifdef PG_TIMELINE        // diagnostic build (make EXTRA=-DPG_TIMELINE): the product's k_solve also records its per-wavefront timeline (costs ~100 B of scratch per lane)
#define PG_TL true
#else
#define PG_TL PROF
#endif
#ifndef PG_SETTLE_PASSES
#define PG_SETTLE_PASSES 4
#endif
// IPM = false: the ROUNDS-ONLY instantiation (round 4): the active-set attempts (previous step's set, empty set) and nothing else -- no interior-point state (t, corr,
// 1/t), no Mehrotra code, no second start.  An instance its rounds do not serve is appended to SolveOut::todo and left untouched; the host launches the full kernel
// over that list behind it (SolveOut::list / n_list: block i solves list[i], blocks beyond *n_list return at once, the attempts already made are skipped).
template <bool PROF, bool RING, bool FUSE, bool IPM = true>
__global__ __launch_bounds__(64, PG_SOLVE_WAVES(RING, IPM)) void k_solve(DevCfg C, int B, real* qp, const real* __restrict__ nodes, SolveOut O, unsigned long long* __restrict__ prof,
                                                                          const tdouble* __restrict__ dt_grid, const real* __restrict__ hji_Mb) {
    // Launch order: slot i of the launch is instance order_in[i] when an order is supplied (filed by the nodes kernels, likely stragglers first: see OrderOut)
    const bool whole = O.mode && *O.mode != 0;                  // (wave-uniform: a scalar load)
    if constexpr (!IPM) { if (whole) return; }                 // the full kernel behind this launch serves the whole batch
    const bool listm = O.n_list && !whole;
    if constexpr (IPM) { if (whole && O.n_whole && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(O.n_whole, 1); }      // (one word per launch: the test of the device-side switch reads it)
    if (listm && (int)blockIdx.x >= *O.n_list) return;         // list mode: nothing left for this block
    // (Round 5, measured and removed: a PERSISTENT grid -- one block per wavefront slot, further instances pulled from a counter -- to save the 4.4 us between the exit of a
    //  wavefront and the entry of the next one on its SIMD.  Same launch time with and without (0.319 / 0.319 ms at 4096 instances): the gap is the prologue's first round trip to
    //  memory, not the dispatch; and the loop-carried register pressure cost the kernel 4 % and the full instantiation 80 B of scratch.  EXPERIMENTS.md 11.)
    unsigned long long pc[6] = {0, 0, 0, 0, 0, 0}, tprev = 0;
    auto stamp = [&](int slot_) { if (PROF) { unsigned long long now = clock64(); pc[slot_] += now - tprev; tprev = now; } };
    if (PROF) tprev = clock64();
    const unsigned long long t_entry = (PG_TL && prof) ? wall_clock64() : 0ull;
    const int b = listm ? O.list[blockIdx.x] : (O.order_in ? O.order_in[blockIdx.x] : (int)blockIdx.x), lane = threadIdx.x;
    const int N = C.N, NN = C.NN;
    // -DPG_TIMELINE build, product instantiation: wall-clock marks (10 ns units since entry) of the FIRST time each point is passed, in the slots the cycle counters of the
    // diagnostic instantiation use: 0 prologue done, 1 first stage assembly done, 2 first matrix pass done, 3 first roll-out done, 4 first check done, 5 epilogue starts
    unsigned tl_seen = 0u;
    auto tl_mark = [&](int slot) __attribute__((always_inline)) {
        if constexpr (PG_TL && !PROF) { if (prof && lane == 0 && !((tl_seen >> slot) & 1u)) { prof[(size_t)b * 6 + slot] = wall_clock64() - t_entry; tl_seen |= 1u << slot; } }
    };
    // FUSE: update_QP! of this instance first, by the wave that is about to solve it -- lane pair (2t, 2t+1) linearises interval t (2N <= 64; lanes beyond 2N mirror
    // the last interval and store nothing).  The QP data go to memory exactly as k_linearize writes them (pg_get_qp reads them; the solve below reads them back
    // through L2: fence + barrier in between).  Why fuse: k_solve ends with its slowest wave, and as a kernel of its own it idles most of the machine during that
    // tail; here the SIMDs that are done with quick instances linearise and solve the next ones meanwhile.
    if constexpr (FUSE) {
        const int lp = lane < 2 * N ? lane : 2 * N - 2 + (lane & 1);
        linearize_pair(C, b, lp >> 1, lp & 1, lane < 2 * N, nodes, dt_grid, hji_Mb, qp);
        __threadfence();
        __syncthreads();
    }
    extern __shared__ real lds[];
    // The dynamics blocks are NOT resident: each pass streams them stage by stage from L2 through a 4-slot LDS ring
    // (slot = SB doubles: rows 0..5 of Abar_k = [A | B0+Bf] at row stride 9, rows 0..5 of Bbar_k = Bf (12), cbar_k = c (6)); rows 6,7 are [0 I] / I / 0.
    real* sRing = lds;                 // RING: [4][SB] slots; otherwise all N stage blocks resident: [N][SB]
    real* sQ = sRing + (RING ? 4 : N) * SB;         // [NN][10]: diagonal[8] (with (Uy,Uy),(r,r) at 2,3), (Uy,r) off-diagonal, (delta,Fx) off-diagonal
    real* sq = sQ + 10 * NN;           // [NN][8]
    real* sR = sq + 8 * NN;            // [N][2]   diagonal of Rhat
    real* sr = sR + 2 * N;             // [N][2]
    real* sK = sr + 2 * N;             // [N][2][8]
    real* sSi = sK + 16 * N;           // [N][4]   (Sinv00, Sinv01, Sinv11, -)
    real* sMc = sSi + 4 * N;           // [N][8]   P_{k+1} cbar_k
    real* skf = sMc + 8 * N;           // [N][2]
    real* sx = skf + 2 * N;            // [NN][8]  Newton point
    real* sv = sx + 8 * NN;            // [N][2]
    real* sP = sv + 2 * N;             // [8][9]   (row stride 9: conflict-free row reads)
    real* sMT = sP + 72;               // [11][9]  (P [Abar | Bbar | cbar]) stored column-major, column stride 9
    real* sx0 = sMT + 99 + 1;              // [8]
    real* sDum = sx0 + 8;              // [72] sink for predicated-off stores (keeps the pass loops branch-free; a lane's slot and the one four further on)
    real* sZero = sDum + 72;           // [2]  a stored 0.0 (off-pattern entries of Qhat)
    real* sOne = sZero + 2;            // [2]  a stored 1.0 (the identity entries of the operand [Abar | Bbar | cbar])
    real* sCst = sOne + 2;             // [32] constant rows of the roll-out: e6 (0..7), e7 (8..15), (1, 0), (0, 1) (16..19), zeros (20..31)
    real* skf_ck = sCst + 32;          // [N][2] the predictor's feed-forward terms as the matrix pass left them (skf itself is rewritten by every vector pass): restored when the recursion restarts at its checkpoint
    real* sF0 = skf_ck + 2 * N;        // [N][11] row 8 of the matrix pass's product C = [Abar Bbar]'[M_A | M_B | y]: F0 = Bbar0' P Abar (8), Bbar0' P Bbar0, S01, Bbar0' y (the last rewritten by a
                                       // vector pass) -- the stationarity condition in the first input, for the multiplier of a pinned rate row; written only while a stage is pinned
    bool any_pin = false;              // (wave-uniform: some stage of this instance is pinned in the round being assembled)

    const QpOff o = qp_offsets(N);
    const real* Q = qp + (size_t)b * C.qp_len;
    // ---- stage blocks come straight from the QP data update_QP! wrote (A[N][36], B0[N][12], Bf[N][12], c[N][6]: no second, packed copy in HBM) and are packed
    // into the LDS layout on the way in: lane l < 36 moves A entry l; lanes 36..47 the (B0, Bf) pair l - 36 (columns 6,7 of Abar are B0 + Bf, Bbar = Bf);
    // lanes 48..53 move c.  RING: two-deep software pipeline (register, then LDS) through a 4-slot ring; otherwise everything lands once ----
    const bool mvA = lane < 36, mvB = lane >= 36 && lane < 48, mvC = lane >= 48 && lane < 54;
    const int src0 = mvA ? o.A + lane : (mvB ? o.B0 + lane - 36 : (mvC ? o.c + lane - 48 : o.A));
    const int sstr = mvA ? 36 : (mvB ? 12 : (mvC ? 6 : 36));
    const int src1 = mvB ? o.Bf + lane - 36 : o.Bf;
    const int dst0 = mvA ? SB_ROW * (lane / 6) + lane % 6 : (mvB ? SB_ROW * ((lane - 36) >> 1) + 6 + ((lane - 36) & 1) : SB_C + lane - 48);
    const int dst1 = SB_B + lane - 36;
    real ring_v0, ring_v1;
    auto ring_slot = [&](int k) -> real* { return sRing + (RING ? (k & 3) : k) * SB; };
    auto ring_load = [&](int k) { int kk = k < 0 ? 0 : (k >= N ? N - 1 : k); ring_v0 = Q[src0 + sstr * kk]; ring_v1 = Q[src1 + 12 * kk]; };
    auto ring_put = [&](int k) {
        real* slot = ring_slot(k);
        *((mvA || mvB || mvC) ? slot + dst0 : sDum + lane) = mvB ? ring_v0 + ring_v1 : ring_v0;
        *(mvB ? slot + dst1 : sDum + lane) = ring_v1;
    };
    // prime(k0, dir): block k0 lands in the ring, block k0+dir is in flight.  step(k, dir) at the top of stage k: block k+dir lands, k+2dir takes off.
    auto ring_prime = [&](int k0, int dir) { if (RING) { ring_load(k0); ring_put(k0); ring_load(k0 + dir); } };
    auto ring_step = [&](int k, int dir) { if (RING) { ring_put(k + dir); ring_load(k + 2 * dir); } };
    // short horizons: every stage block is read from HBM exactly once and stays in LDS (SB N doubles = 17.3 KB at N = 30).  ALL the loads of the prologue -- the N stage
    // blocks (two values per lane and stage), the fixed first node, the per-stage constants below -- are issued as ONE batch before anything waits for any of them
    // (round 4; rounds 1-3 filled the LDS in batches of six stages, each a round trip to the memory side, and loaded the rest in three more dependent round trips)
    constexpr int NFILL = RING ? 1 : 32;
    real fill0[NFILL], fill1[NFILL];
    if (!RING) {
#pragma unroll
        for (int u = 0; u < NFILL; u++) { const int kk = u < N ? u : N - 1; fill0[u] = Q[src0 + sstr * kk]; fill1[u] = Q[src1 + 12 * kk]; }
    }
    const real x0_fill = lane < 6 ? Q[o.qcurr + lane] : Q[o.ucurr + (lane < 8 ? lane - 6 : 0)];
    const int prev_solved = O.solved[b], prev_status = O.status[b];      // (the warm-start test further down: loaded with the batch, not in a round trip of its own)
    const real hji_b = Q[o.b];
    if (lane < 4) sZero[lane] = lane < 2 ? real(0.0) : real(1.0);
    if (lane < 32) sCst[lane] = (lane == 6 || lane == 15 || lane == 16 || lane == 19) ? real(1.0) : real(0.0);

    // ---- per-stage constants in the registers of lane s (stage s = transition s, node s+1) ----
    const bool act = lane < N;
    const int s = act ? lane : 0;
    real h0[4], h1[4], bb[NROW];
    const real M0 = Q[o.M], M1 = Q[o.M + 1];
    const bool hji_on = act && (s + 1 < (C.cp.N_HJI < C.Ns ? C.cp.N_HJI : C.Ns));
    // wall extension (lateral formulation): rows 0, 1, 2 -- bounds on the INERT Ux / Fx slots of the embedding there -- carry  e <= edge_L + sw,  e >= edge_R - sw,
    // sw >= 0, and sw takes the slot of the (absent) safety-row slack: the third stage-locally eliminated slack group
    const bool wall_on = act && C.walls != 0;
    const real dts = Q[o.dt + s];
    const real Rd0 = real(2.0) * C.cp.R_ddelta / dts, Rd1 = real(2.0) * C.cp.R_dFx / dts;
    const real wb = C.cp.W_beta * dts, wr = C.cp.W_r * dts, wh = C.cp.W_HJI;
#pragma unroll
    for (int i = 0; i < 4; i++) { h0[i] = Q[o.H + 8 * s + 2 * i]; h1[i] = Q[o.H + 8 * s + 2 * i + 1]; bb[6 + i] = Q[o.G + 4 * s + i]; }
    bb[0] = -C.cp.V_min; bb[1] = C.cp.V_max; bb[2] = -C.fxmin_n; bb[3] = Q[o.dmax + s]; bb[4] = -Q[o.dmin + s]; bb[5] = Q[o.fxmax + s];
    bb[10] = real(0.0); bb[11] = real(0.0); bb[12] = Q[o.ddmax + s]; bb[13] = -Q[o.ddmin + s]; bb[14] = Q[o.b]; bb[15] = real(0.0);
    if (wall_on) { const real* w = C.wall_edges + ((size_t)b * N + s) * 2; bb[0] = w[0]; bb[1] = -w[1]; bb[2] = real(0.0); }
    const real ww = C.wall_weight * dts;
    const real Qd5 = real(2.0) * C.cp.Q_e * dts;
    if (!RING) {      // (the loads issued at the top have had the whole constant set-up to arrive)
#pragma unroll
        for (int u = 0; u < NFILL; u++) {
            real* slot = ring_slot(u < N ? u : N - 1);
            *(((mvA || mvB || mvC) && u < N) ? slot + dst0 : sDum + lane) = mvB ? fill0[u] + fill1[u] : fill0[u];
            *((mvB && u < N) ? slot + dst1 : sDum + lane) = fill1[u];
        }
    }
    if (lane < 8) sx0[lane] = x0_fill;
    if (act) {   // entries of the stage cost that never change
        real* Qo = sQ + 10 * (s + 1);
        Qo[0] = real(2.0) * C.cp.Q_ds * dts; Qo[4] = real(2.0) * C.cp.Q_dpsi * dts; Qo[5] = real(2.0) * C.cp.Q_e * dts;
        real* qo = sq + 8 * (s + 1);
        qo[0] = real(0.0); qo[4] = real(0.0); qo[5] = real(0.0);
        sR[2 * s + 1] = Rd1; sr[2 * s + 1] = real(0.0);
    }
    const real Qd6 = real(2.0) * C.cp.R_delta * dts, Qd7 = real(2.0) * C.cp.R_Fx * dts;
    __syncthreads();

    // slack of every row at the point w = (x[8], v0, s1, s2, sh)
    auto slacks = [&](const real* x, real v0, real s1, real s2, real sh, real* out) {
        out[0] = wall_on ? bb[0] - x[5] + sh : x[1] + bb[0]; out[1] = wall_on ? bb[1] + x[5] + sh : bb[1] - x[1]; out[2] = wall_on ? sh : x[7] + bb[2];
        out[3] = bb[3] - x[6]; out[4] = x[6] + bb[4]; out[5] = bb[5] - x[7];
#pragma unroll
        for (int i = 0; i < 4; i++) out[6 + i] = bb[6 + i] - (h0[i] * x[2] + h1[i] * x[3]) + (i < 2 ? s1 : s2);
        out[10] = s1; out[11] = s2; out[12] = bb[12] - v0; out[13] = v0 + bb[13];
        out[14] = bb[14] + M0 * x[6] + M1 * x[7] + sh; out[15] = sh;
    };
    const int nrows = hji_on ? 16 : 14;
    // Rows that can never be active: delta at node s+1 cannot reach its bound when even the rate limit on every transition up to s leaves it short of it
    // (delta_{s+1} <= delta_0 + sum_{k<=s} ddmax_k).  The polish never takes such a row into its working set: at a point that overshoots -- the optimum of a
    // working set that still lacks some rate rows does -- the bound looks violated, and holding it as an equality TOGETHER with the rate rows before it is an
    // inconsistent system (the typical failure of the add-all-violated rule on a rate-limited ramp of the steering angle to its stop).
    // The mirror image: a working set that holds the rate row on EVERY transition up to s puts delta_{s+1} exactly at that reach; where the reach lies beyond the
    // steering bound the rate row of transition s cannot be part of such a run (`overshoot`; the polish takes it out before it solves: the previous step's set,
    // shifted by one stage on the new QP, is the typical offender -- a ramp to the stop that now ends one stage earlier).
    unsigned addable = 0xFFFFu, overshoot = 0u, excl = 0u;
    {
        real up = act ? bb[12] : real(0.0), dn = act ? bb[13] : real(0.0);
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { real a = __shfl_up(up, d), c = __shfl_up(dn, d); if (lane >= d) { up += a; dn += c; } }
        const real d0 = sx0[6], slk = C.polish_tol * real(10.0);
        if (d0 + up < bb[3] - slk) addable &= ~(1u << 3);
        if (d0 - dn > -bb[4] + slk) addable &= ~(1u << 4);
        if (act && d0 + up > bb[3] + slk) overshoot |= 1u << 12;
        if (act && d0 - dn < -bb[4] - slk) overshoot |= 1u << 13;
        // Pairs of delta rows that cannot be active together (delta is an integrator chain with box and rate limits; dmax_p / dmin_p = the bounds of delta at the
        // START of this lane's transition, node 0 being the fixed delta_0):
        //   bit 0: delta_{s+1} at its upper bound AND the rate row that steers down at full rate: delta_s would sit above its own upper bound;  bit 1: the mirror image;
        //   bit 2: delta_s and delta_{s+1} both at their upper bounds leave the increment no room to be at either rate limit;                   bit 3: the mirror image.
        // A row whose partner is in the working set (or joins it in the same round) is not added: at the optimum of a set that lacks a row, delta overshoots around
        // the missing row and these partners LOOK violated (the row that ends a ramp and the first row of the stop are the typical pair).
        const real dmax_p = lane > 0 ? __shfl_up(bb[3], 1) : d0, dmin_p = lane > 0 ? -__shfl_up(bb[4], 1) : d0, ddm = fmin(bb[12], bb[13]);
        if (bb[3] + bb[13] > dmax_p + slk) excl |= 1u;
        if (-bb[4] - bb[12] < dmin_p - slk) excl |= 2u;
        if (lane > 0 && fabs(bb[3] - dmax_p) < ddm - slk) excl |= 4u;
        if (lane > 0 && fabs(-bb[4] - dmin_p) < ddm - slk) excl |= 8u;
    }
    const int r8 = lane & 7;             // component index for the vector passes (lanes >= 8 mirror lanes 0..7)

    // forward roll-out: lanes r8 hold x_k[r8] in a register; broadcasts by readlane; results published to sx/sv for the stage lanes.
    // The loop body is one basic block (predicated stores go to a dummy slot) so that all LDS reads of a stage issue back to back.
    // Lane roles inside each 16-lane row (rows mirror each other): lanes 0..5 own the six dynamics rows of Abar, lanes 6, 7 the two input states, lanes 8, 9 the two
    // rows of the gain K -- every lane does ONE 8-term dot product against the broadcast state and the two inputs come back through SGPRs (round 1 had all eight
    // component lanes evaluate both gain rows as well: 24 FMAs and 29 LDS reads per lane and stage, now 10 and 11).
    // Working-set guess of a cold instance (round 4).  The verified sets of a cold batch are runs of steering-rate rows from the first transition (7 rows for the instances that
    // used to take two rounds, 10 for three, 13-15 beyond: the rounds followed the length of the run, one row per round past the second).  The roll-out of the FIRST round --
    // empty set: the unconstrained optimum -- therefore applies the gains with the steering rate clipped at its limits: the lag of the held transitions feeds back through
    // the gains and the later transitions saturate as they will in the optimum.  Nothing clipped: the point is the unconstrained optimum and verifies as before (half of the
    // batch).  Otherwise the point is not a KKT candidate; its clipped transitions are the first working set (no multiplier update, no drops in that check).
    // Headline batch: rounds 1 / 2 / 3 / 4+ 2105 / 714 / 928 / 349 -> 2105 / 1552 / 372 / 67 instances (mean 1.97 -> 1.62), k_solve 0.386 -> 0.334 ms.  Measured and dropped:
    // clipping in every round (11 instances end in the interior point), clipping at the steering stops as well (fp64: 3-round instances 372 -> 418, 4+ 67 -> 21, same time; fp32: more 9-12 round instances, 0.34 -> 0.43 ms; and the code alone costs 4 % in the roll-out).  Option "clip_guess" = 0: off.
    bool clip_now = false, clip_used = false, clip_off = false; unsigned clip_mask = 0u;
    // Held steering-RATE rows (12, 13) are eliminated EXACTLY (round 5): such a row pins the first input of its stage, v0 = +ddmax / -ddmin.  The stage cost of a pinned stage
    // carries  Rhat0 = BIGP, rhat0 = -BIGP v0  -- a penalty so large that 1 / BIGP vanishes against everything else in the arithmetic (1e200; 1e22 in fp32): the UNCHANGED
    // recursion then returns K0 = 0, kff0 = v0 and (K1, kff1) from the 1 x 1 pivot S11 with the cross term S01 v0 in its right-hand side to the last bit -- no select on the
    // serial chain, nothing added to the roll-out.  The row's multiplier is read off the stationarity condition in v0 by the stage's own lane behind the roll-out:
    //   lambda = -/+ (F0 x + Bbar0'y + S00 v0 + S01 v1),  F0, Bbar0'P Bbar0, S01, Bbar0'y left in LDS by the matrix (vector) pass; in a correction pass x, v are the corrections
    //   and the first input's own gradient is Rhat0 v0 (the re-centred problem's multiplier is the multiplier).
    // No penalty iteration, t = 0 exactly: a working set of rate rows and slack pivots (every two- and three-round instance of a cold tracking batch) verifies at its first
    // check, without the refinement pass the augmented Lagrangian needed (vector pass + roll-out: 13 of the ~37 us of such an instance's last round).
    // (First version of this round: K0 = 0 / kff0 = v0 / 1 x 1 pivot through selects in the matrix pass and the multiplier formed inside the roll-out -- +150 cycles per stage
    //  of the matrix pass, +10 % per round, and the warm steps of a closed loop, one round each with nothing to save, went from 0.59 to 0.62 ms.)
    unsigned amask = 0;                // (this stage's rows currently held active -- the polish state further down)
    const real BIGP = sizeof(real) == 8 ? real(1e200) : real(1e22);
#if !defined(PG_NO_MFMA)
    constexpr bool EXR = true;
#else
    constexpr bool EXR = false;        // (the VALU fall-back of the matrix pass keeps the penalty form)
#endif
    bool exr = EXR;                    // (per instance, i.e. wave-uniform: see where the safety row is examined)
    auto forward = [&](auto use_gain_t, bool delta = false) {      // delta: the roll-out of a CORRECTION (starts at 0, no affine term: see the polish refinement)
        constexpr bool use_gain = decltype(use_gain_t)::value;
        // Every lane evaluates x+ = row . x + addc + bf . v with ITS row, constant and input column, each an LDS address and a stride per stage fixed here (the loop is bound by
        // instruction issue: no value selects, no blend): lanes 0..5 the six dynamics rows of [Abar | cbar | Bbar]; lanes 6, 7 the input states as the rows e6, e7 with the
        // input columns (1, 0), (0, 1) from a stored table; lanes 8, 9 the gain rows with kff (v = 0 roll-out: zeros); the rest zeros.  Rows of 16 lanes mirror each other.
        const int f16 = lane & 15;
        const bool isA = f16 < 6, isK = use_gain && (f16 == 8 || f16 == 9), isU = f16 == 6 || f16 == 7;
        const real* const zrow = sCst + 20;
        const real* const prow = isA ? sRing + SB_ROW * f16 : (isU ? sCst + 8 * (f16 - 6) : (isK ? sK + 8 * (f16 - 8) : zrow));
        const int trow = isA ? SB : (isK ? 16 : 0);
        const real* const padd = (isA && !delta) ? sRing + SB_C + f16 : (isK ? skf + (f16 - 8) : zrow);
        const int tadd = (isA && !delta) ? SB : (isK ? 2 : 0);
        const real* const pbf = isA ? sRing + SB_B + 2 * f16 : (isU ? sCst + 16 + 2 * (f16 - 6) : zrow);
        const int tbf = isA ? SB : 0;
        real* const dX = lane < 8 ? sx + 8 + lane : sDum + lane;   const int tX = lane < 8 ? 8 : 0;
        real* const dV = lane < 2 ? sv + lane : sDum + lane;       const int tV = lane < 2 ? 2 : 0;
        real xi = delta ? real(0.0) : sx0[r8];
        *(lane < 8 ? sx + lane : sDum + lane) = xi;
        ring_prime(0, +1);
        // (measured and dropped: the row, constant and input column of stage k + 1 loaded while stage k computes -- eleven more live values: 0.2305 -> 0.232 ms)
#pragma unroll 1
        for (int k = 0; k < N; k++) {
            ring_step(k, +1);
            const real* Rk = ring_slot(k);
            // (RING: the stage block sits in slot k & 3 of the ring -- the lanes that read it form their address per stage)
            const real* rowp = RING ? (isA ? Rk + SB_ROW * f16 : prow + trow * k) : prow + trow * k;
            real rw[8];
#pragma unroll
            for (int m = 0; m < 8; m++) rw[m] = rowp[m];
            const real addc = RING ? *((isA && !delta) ? Rk + SB_C + f16 : padd + tadd * k) : padd[tadd * k];
            const real* bfp = RING ? (isA ? Rk + SB_B + 2 * f16 : pbf) : pbf + tbf * k;
            const real bf0 = bfp[0], bf1 = bfp[1];
            real xm[8];      // (measured and dropped: x_k[m] broadcast inside the multiply-add, v_fmac_f64_dpp row_newbcast -- 8 instructions for 24 --: 0.2305 against 0.232 ms, in the noise)
#pragma unroll
            for (int m = 0; m < 8; m++) xm[m] = rl(xi, m);
            real d0 = addc, d1 = real(0.0);
#pragma unroll
            for (int m = 0; m < 8; m += 2) { d0 += rw[m] * xm[m]; d1 += rw[m + 1] * xm[m + 1]; }
            const real d = d0 + d1;
            real v0 = rl(d, 8); const real v1 = rl(d, 9);
            if (clip_now) {        // (wave-uniform; first round of a cold instance) saturated roll-out: the gain of the unconstrained problem, the steering rate held inside its limits
                const real hi = rl(bb[12], k), lo = -rl(bb[13], k);
                unsigned cm = 0u;
                if (v0 > hi) { v0 = hi; cm = 1u << 12; }
                else if (v0 < lo) { v0 = lo; cm = 1u << 13; }
                if (C.clip_stops != 0) {      // ... and the steering angle inside its stops: a ramp that reaches the stop ends there, and the stop's rows are part of the guess
                    const real dn = xm[6] + v0, smax = rl(bb[3], k), smin = -rl(bb[4], k);
                    if (dn > smax) { v0 = smax - xm[6]; cm = (v0 < hi ? 0u : cm) | (1u << 3); }
                    else if (dn < smin) { v0 = smin - xm[6]; cm = (v0 > lo ? 0u : cm) | (1u << 4); }
                }
                if (lane == k) clip_mask |= cm;
            }
            const real xn = d + (bf0 * v0 + bf1 * v1);        // (lanes 8..15 of a row: not a state, never read)
            xi = xn;
            dX[tX * k] = xn;
            dV[tV * k] = lane == 0 ? v0 : v1;
        }
        __syncthreads();
    };

    StageRows R;
    // the damped iterate (x_{s+1}, sigma) of this stage is kept in the output buffers (read-modify-write once per iteration), not in registers
    real* const SXs = O.sol_x + (size_t)b * NN * 8 + 8 * (s + 1);
    real* const SGs = O.sol_sigma + ((size_t)b * N + s) * 3;
    real rp0 = real(0.0);
    const real ntot = wave_sum(act ? (real)nrows : real(0.0));
    real phi = real(1.0), mu = real(0.0);
    int it = 0, status = PG_MAX_ITER;

    real e_d1, e_c10, e_c11, e_g1, e_d2, e_c20, e_c21, e_g2, e_dh, e_ch0, e_ch1, e_gh;   // e_d* hold RECIPROCALS of the slack pivots
    real it_[NROW];                    // 1 / t_j, refreshed once per interior-point iteration
    // Polish state (see the loop below): pmode = 0 while the interior point runs, then the round number of the active-set polish;
    // amask = this stage's rows currently held active.
    int pmode = 0, pstat = 0; unsigned mask_ipm = 0, amask_1ago = 0xFFFFFFFFu, amask_2ago = 0xFFFFFFFFu; bool polish_gave_up = false, cycle_broken = false, settle_used = false; int settle_left = 0;
    const real rho = C.polish_rho, ptol = C.polish_tol, dtol = real(1000.0) * C.polish_tol;      // dtol: largest correction of the last refinement pass a verified point may have had
    auto assemble = [&](real sigmu, bool matrices) {
        real W[NROW], ell[NROW];
        if (!IPM || pmode) {
            // polish: active rows are equalities enforced by the augmented Lagrangian  -y t(z) + rho/2 t(z)^2  (y lives in R.lam), inactive rows are absent.
            // Same shape as the barrier terms: W = rho, constant part of the multiplier = y - rho b.
#pragma unroll
            for (int j = 0; j < NROW; j++) {
                const bool a = j < nrows && ((amask >> j) & 1u) && !(exr && (j == 12 || j == 13));      // (held rate rows: eliminated exactly in the recursion, not penalised)
                W[j] = a ? rho : real(0.0);
                ell[j] = a ? R.lam[j] - rho * bb[j] : real(0.0);
            }
        } else if constexpr (IPM) {
#pragma unroll
            for (int j = 0; j < NROW; j++) {
                bool on = j < nrows;
                W[j] = on ? R.lam[j] * it_[j] : real(0.0);
                ell[j] = on ? (sigmu - R.corr[j]) * it_[j] + R.lam[j] - W[j] * bb[j] : real(0.0);
            }
        }
        real g1 = wall_on ? real(0.0) : -ell[0] + ell[1], g7 = (wall_on ? real(0.0) : -ell[2]) + ell[5] - M1 * ell[14], g6 = ell[3] - ell[4] - M0 * ell[14];
        real g2 = real(0.0), g3 = real(0.0);
#pragma unroll
        for (int i = 0; i < 4; i++) { g2 += h0[i] * ell[6 + i]; g3 += h1[i] * ell[6 + i]; }
        e_g1 = wb - ell[6] - ell[7] - ell[10]; e_g2 = wr - ell[8] - ell[9] - ell[11]; e_gh = wall_on ? ww - ell[0] - ell[1] - ell[2] : wh - ell[14] - ell[15];
        real gv0 = ell[12] - ell[13];
        e_d1 = frcp(W[6] + W[7] + W[10]); e_d2 = frcp(W[8] + W[9] + W[11]); e_dh = wall_on ? frcp(W[0] + W[1] + W[2]) : (hji_on ? frcp(W[14] + W[15]) : real(1.0));
        e_c10 = -(W[6] * h0[0] + W[7] * h0[1]); e_c11 = -(W[6] * h1[0] + W[7] * h1[1]);
        e_c20 = -(W[8] * h0[2] + W[9] * h0[3]); e_c21 = -(W[8] * h1[2] + W[9] * h1[3]);
        e_ch0 = wall_on ? -(W[0] - W[1]) : W[14] * M0; e_ch1 = wall_on ? real(0.0) : W[14] * M1;      // wall rows: t = b -/+ e + sw, i.e. the envelope-row pattern with h = +1, -1 on e
        if (!hji_on && !wall_on) e_gh = real(0.0);
        any_pin = __any(act && exr && (pmode != 0 || !IPM) && (amask & 0x3000u) != 0u);
        if (act) {
            real* qo = sq + 8 * (s + 1);
            qo[1] = g1;
            qo[2] = g2 - e_c10 * e_g1 * e_d1 - e_c20 * e_g2 * e_d2;
            qo[3] = g3 - e_c11 * e_g1 * e_d1 - e_c21 * e_g2 * e_d2;
            qo[6] = g6 - (wall_on ? real(0.0) : e_ch0 * e_gh * e_dh); qo[7] = g7 - e_ch1 * e_gh * e_dh;
            if (wall_on) qo[5] = (ell[0] - ell[1]) - e_ch0 * e_gh * e_dh;
            const bool pinned = exr && (pmode != 0 || !IPM) && (amask & 0x3000u) != 0u;
            const real vpin = (amask & (1u << 12)) ? bb[12] : -bb[13];
            sr[2 * s] = pinned ? -BIGP * vpin : gv0;
            if (matrices) {
                real* Qo = sQ + 10 * (s + 1);
                if (wall_on) Qo[5] = Qd5 + W[0] + W[1] - e_ch0 * e_ch0 * e_dh;
                Qo[1] = wall_on ? real(0.0) : W[0] + W[1];
                Qo[6] = Qd6 + W[3] + W[4] + M0 * M0 * W[14] - (wall_on ? real(0.0) : e_ch0 * e_ch0 * e_dh);
                Qo[7] = Qd7 + (wall_on ? real(0.0) : W[2]) + W[5] + M1 * M1 * W[14] - e_ch1 * e_ch1 * e_dh;
                real yy = real(0.0), yr = real(0.0), rr = real(0.0);
#pragma unroll
                for (int i = 0; i < 4; i++) { yy += W[6 + i] * h0[i] * h0[i]; yr += W[6 + i] * h0[i] * h1[i]; rr += W[6 + i] * h1[i] * h1[i]; }
                Qo[2] = yy - e_c10 * e_c10 * e_d1 - e_c20 * e_c20 * e_d2;
                Qo[8] = yr - e_c10 * e_c11 * e_d1 - e_c20 * e_c21 * e_d2;
                Qo[3] = rr - e_c11 * e_c11 * e_d1 - e_c21 * e_c21 * e_d2;
                Qo[9] = M0 * M1 * W[14] - e_ch0 * e_ch1 * e_dh;
                sR[2 * s] = pinned ? BIGP : Rd0 + W[12] + W[13];
            }
        }
    };
    const int li = lane >> 3, lj = lane & 7;
    // slot of Qhat[li][lj] inside the packed 10-double node record; off-pattern entries read a stored zero (qmul = 0 kills the node stride)
    const int qidx = (li == lj) ? li : (((li == 2 && lj == 3) || (li == 3 && lj == 2)) ? 8 : (((li == 6 && lj == 7) || (li == 7 && lj == 6)) ? 9 : -1));
    const real* qbase = qidx >= 0 ? sQ + qidx : sZero;
    const int qmul = qidx >= 0 ? 10 : 0;

#if !defined(PG_NO_MFMA)
    // ---- Riccati matrix pass on the matrix cores: v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32, lane = (g, c) = (lane >> 4, lane & 15) ------------------------
    // The stage recursion  M = P [A B c],  C = [A B]' [M_A M_B y],  P_k = Q + A'M_A + F'K,  K = -S^-1 F  is three chained products of genuinely dense 8 x {8,11}
    // blocks.  Operand layout of the instruction: A-operand lane holds A[m = c][k = 4s + g], B-operand lane holds B[k = 4s + g][n = c].  In the "operand layout"
    // used below a matrix Z is held as two values per lane, z_s = Z[4s + g][c] (s = 0, 1: rows 0..7): that IS the B operand of k-step s, and -- P being
    // symmetric -- also the A operand.  The fp64 instruction returns result register r = D[g + 4r][c], i.e. registers 0, 1 of a RESULT are already in operand
    // layout: the chain needs no cross-lane movement at all; the fp32 instruction returns D[4g + r][c] and needs one 4 x 4 (row group x register) transpose per
    // hand-over, four v_permlane{16,32}_swap.  The 8 x 8 x 11 product that cost 24 LDS reads + 14 FMAs per lane on the VALU costs two operand reads and two
    // MFMAs.  Column 10 carries the predictor's VECTOR recursion through the same products: M[:,10] = P c, the second product is fed y = P c + p_{k+1} in
    // that column, and the third adds F' kff, so that column 10 of the result is A'y + K'f = p_k - q.
    // Round 1 ran this pass on the VALU with two LDS exchanges per stage (46 % of the kernel: 2450 cycles per stage, co-limited by LDS bandwidth and issue).
    const int mg = lane >> 4, mc = lane & 15;
    // Every per-lane choice of the stage loop below is made ONCE, here, as an address and a stride per stage (LDS addresses are one dword: a select on an address is one
    // instruction, a select on an fp64 value two; the loop is bound by instruction issue -- EXPERIMENTS.md 11.8): operands that are constants for some lanes come from a stored
    // 0.0 / 1.0 with stride 0, predicated-off stores go to the lane's dummy slot with stride 0, one-hot lane weights replace value selects.
    int xoff[2]; bool xlds[2]; const real* xsrc[2]; int xstr[2];
#pragma unroll
    for (int sgi = 0; sgi < 2; sgi++) {            // operand X[4 s + g][c] of X = [Abar | Bbar | cbar] (8 x 11): rows 0..5 live in the LDS stage block, rows 6, 7 are [0 I | I | 0]
        const int row = 4 * sgi + mg;
        xlds[sgi] = row < 6 && mc < 11;
        xoff[sgi] = !xlds[sgi] ? 0 : (mc < 8 ? SB_ROW * row + mc : (mc < 10 ? SB_B + 2 * row + (mc - 8) : SB_C + row));
        const bool one = (row == 6 && (mc == 6 || mc == 8)) || (row == 7 && (mc == 7 || mc == 9));
        xsrc[sgi] = xlds[sgi] ? sRing + xoff[sgi] : (one ? sOne : sZero);
        xstr[sgi] = xlds[sgi] ? SB : 0;
    }
    const real m10 = mc == 10 ? real(1.0) : real(0.0), w0 = mg == 0 ? real(1.0) : real(0.0), w1 = mg == 1 ? real(1.0) : real(0.0);
    real* const dMc = mc == 10 ? sMc + mg : sDum + lane;                       const int tMc = mc == 10 ? 8 : 0;       // Mc_k: rows mg and mg + 4
    real* const dKf = mg < 2 ? (mc < 8 ? sK + 8 * mg + mc : (mc == 10 ? skf + mg : sDum + lane)) : sDum + lane;
    const int tKf = mg < 2 ? (mc < 8 ? 16 : (mc == 10 ? 2 : 0)) : 0;                                                    // gains (columns 0..7) and feed-forward (column 10): one store
    real* const dSi = lane < 3 ? sSi + lane : sDum + lane;                     const int tSi = lane < 3 ? 4 : 0;
    const real* abase[2]; int amul[2];             // what is added to the operand-layout value s: Qhat_k[4s + g][c] (c < 8), qhat_k[4s + g] (c == 10), nothing elsewhere
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const int row = 4 * r + mg;
        const int qi = (row == mc) ? row : (((row == 2 && mc == 3) || (row == 3 && mc == 2)) ? 8 : (((row == 6 && mc == 7) || (row == 7 && mc == 6)) ? 9 : -1));
        abase[r] = mc < 8 ? (qi >= 0 ? sQ + qi : sZero) : (mc == 10 ? sq + row : sZero);
        amul[r] = mc < 8 ? (qi >= 0 ? 10 : 0) : (mc == 10 ? 8 : 0);
    }
#ifdef PG_F32
    typedef float mfma_acc __attribute__((ext_vector_type(4)));
#define PG_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)
    // result (register r of lane (g, c) = D[4g + r][c]) -> operand layout: o_s = D[4s + g][c] = register g of lane (s, c): the 4 x 4 transpose of (row group, register)
    auto to_operands = [&](const mfma_acc& a, real& o0, real& o1) {
        auto p01 = __builtin_amdgcn_permlane16_swap(__float_as_uint(a[0]), __float_as_uint(a[1]), false, false);      // rows: (R0.0, R1.0, R0.2, R1.2) / (R0.1, R1.1, R0.3, R1.3)
        auto p23 = __builtin_amdgcn_permlane16_swap(__float_as_uint(a[2]), __float_as_uint(a[3]), false, false);
        auto q0 = __builtin_amdgcn_permlane32_swap(p01[0], p23[0], false, false);                                       // [0]: (R0.0, R1.0, R2.0, R3.0)
        auto q1 = __builtin_amdgcn_permlane32_swap(p01[1], p23[1], false, false);                                       // [0]: (R0.1, R1.1, R2.1, R3.1)
        o0 = __uint_as_float(q0[0]); o1 = __uint_as_float(q1[0]);
    };
    // rows 8, 9 of a result (D[8 + j][c] = register j of lane (2, c)) broadcast to every row group
    auto rows89 = [&](const mfma_acc& a, real& f0, real& f1) {
        auto bc = [&](float v) { unsigned u = __float_as_uint(v); auto s2 = __builtin_amdgcn_permlane32_swap(u, u, false, false);      // [1]: rows (r2, r3, r2, r3)
                                 auto t2 = __builtin_amdgcn_permlane16_swap(s2[1], s2[1], false, false); return __uint_as_float(t2[0]); };   // [0]: (r2, r2, r2, r2)
        f0 = bc(a[0]); f1 = bc(a[1]);
    };
#else
    typedef double mfma_acc __attribute__((ext_vector_type(4)));
#define PG_MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0)
    auto to_operands = [&](const mfma_acc& a, real& o0, real& o1) { o0 = a[0]; o1 = a[1]; };          // register r = D[g + 4r][c]: already the operand of k-step r
    // rows 8, 9 (register 2 of lanes (0, c), (1, c)) broadcast to every row group: two lane-half / row swaps per dword, no LDS
    auto rows89 = [&](const mfma_acc& a, real& f0, real& f1) {
        unsigned lo = (unsigned)__double2loint(a[2]), hi = (unsigned)__double2hiint(a[2]);
        auto sl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false); auto sh = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);      // [0]: rows (r0, r1, r0, r1)
        auto tl = __builtin_amdgcn_permlane16_swap(sl[0], sl[0], false, false); auto th = __builtin_amdgcn_permlane16_swap(sh[0], sh[0], false, false);
        f0 = __hiloint2double((int)th[0], (int)tl[0]); f1 = __hiloint2double((int)th[1], (int)tl[1]);
    };
#endif
    // Checkpoint of the recursion (rounds of the active-set polish, rounds-only instantiation, N <= 32): (P, p) after stage ck_k, two registers per lane.  The stage costs
    // of a round differ from the previous round's only where the working set does, and the sets of a cold batch are runs of rows from the first transition (stage <= 14
    // of 30): while no stage lane >= ck_k - 1 holds anything but its slack pivots -- now and when the checkpoint was taken -- the recursion restarts there (the gains and the
    // Mc of the stages behind it are still in LDS).  The pivots' multipliers are their known values (they move at rounding level once the refinement passes run).
    real ck_V0 = real(0.0), ck_V1 = real(0.0); int ck_top = 1 << 20;      // ck_top: highest stage with a non-pivot row when the checkpoint was taken (1 << 20: none taken)
    const int ck_k = ((N >> 1) + 3) & ~3;
    // (fp64 only: in fp32 the pivots' multipliers drift by 1e-7 per refinement pass, the checkpoint goes stale, and 75 of 8192 instances of the config-4 batch ended in the
    //  interior point)
    constexpr bool CKPT = !IPM && !RING && sizeof(real) == 8;
    bool ck_restart = false;
    auto riccati_matrices = [&]() {
        const bool restart = CKPT && ck_restart;
        real V0 = restart ? ck_V0 : abase[0][amul[0] * N], V1 = restart ? ck_V1 : abase[1][amul[1] * N];          // terminal: P_N = Qhat_N, p_N = qhat_N (operand layout; the vector lives in column 10)
        ring_prime(N - 1, -1);
        if (restart) { const int i = 2 * ck_k + lane; if (i < 2 * N) skf[i] = skf_ck[i]; }      // (2 (N - ck_k) <= 64 values)
        // two segments of the same rolled loop: stages N-1 .. ck_k (skipped on a restart), then ck_k-1 .. 0; the checkpoint is taken between them
        const int nseg = CKPT ? 2 : 1;
#pragma unroll 1
        for (int seg = restart ? 1 : 0; seg < nseg; seg++) {
        const int k_hi = (nseg == 2 && seg == 1) ? ck_k - 1 : N - 1, k_lo = (nseg == 2 && seg == 0) ? ck_k : 0;
        // stage constants (none of these reads depends on the recursion).  PREF: those of stage k - 1 are loaded while the first product of stage k runs -- eight more live
        // values: the fp32 full kernel has the registers (config 3's solve launch 0.595 -> 0.551 ms); the fp64 rounds-only kernel does not (0.233 -> 0.240 ms: more moves
        // through the accumulation registers), and the fp32 rounds-only kernel would leave its two waves per SIMD or spill
        constexpr bool PREF = !RING && IPM && sizeof(real) == 4;
        real nb10 = real(0.0), nb11 = real(0.0), nadd0 = real(0.0), nadd1 = real(0.0), nR0 = real(0.0), nR1 = real(0.0), nr0 = real(0.0), nr1 = real(0.0);
        auto stage_constants = [&](int kk) __attribute__((always_inline)) {
            nb10 = xsrc[0][xstr[0] * kk]; nb11 = xsrc[1][xstr[1] * kk]; nadd0 = abase[0][amul[0] * kk]; nadd1 = abase[1][amul[1] * kk];
            nR0 = sR[2 * kk]; nR1 = sR[2 * kk + 1]; nr0 = sr[2 * kk]; nr1 = sr[2 * kk + 1];
        };
        if (PREF) stage_constants(k_hi);
#pragma unroll 1
        for (int k = k_hi; k >= k_lo; k--) {
            ring_step(k, -1);
            const real* Xk = ring_slot(k);
            real b10, b11, add0, add1, R0, R1, r0v, r1v;
            if (!PREF) {
                b10 = RING ? *(xlds[0] ? Xk + xoff[0] : xsrc[0]) : xsrc[0][xstr[0] * k]; b11 = RING ? *(xlds[1] ? Xk + xoff[1] : xsrc[1]) : xsrc[1][xstr[1] * k];
                add0 = abase[0][amul[0] * k]; add1 = abase[1][amul[1] * k];
                R0 = sR[2 * k]; R1 = sR[2 * k + 1]; r0v = sr[2 * k]; r1v = sr[2 * k + 1];
            } else { b10 = nb10; b11 = nb11; add0 = nadd0; add1 = nadd1; R0 = nR0; R1 = nR1; r0v = nr0; r1v = nr1; }
            // M = P X
            mfma_acc Ma = {real(0.0), real(0.0), real(0.0), real(0.0)};
            // (A operand = P through its own result: lane (g, c) supplies row c of the product.  Columns 8 .. 15 of V hold other things -- S, the vector recursion -- and
            //  give rows 8 .. 15 of M, which nothing reads: no masking.  The same holds for row 10 of C below, the cbar column of X.)
            Ma = PG_MFMA(V0, b10, Ma);
            Ma = PG_MFMA(V1, b11, Ma);
            if (PREF) stage_constants(k > 0 ? k - 1 : 0);      // (in the shadow of the two products)
            real M0, M1; to_operands(Ma, M0, M1);
            { real* d = dMc + tMc * k; d[0] = M0; d[4] = M1; }                       // Mc_k = P_{k+1} cbar_k for the corrector's vector pass
            // C = [Abar Bbar]' [M_A | M_B | y]   (y = P c + p in column 10)
            mfma_acc Cc = {real(0.0), real(0.0), real(0.0), real(0.0)};
            Cc = PG_MFMA(b10, fma(V0, m10, M0), Cc);
            Cc = PG_MFMA(b11, fma(V1, m10, M1), Cc);
            // rows 8, 9 of C: F = Bbar' M_A (c < 8), S - Rhat (c = 8, 9), Bbar' y (c = 10).
            // The pivot's three numbers are read straight from the result registers (fp64: D[8][c] = register 2 of lane c, D[9][c] = register 2 of lane 16 + c; fp32: D[8 + j][c]
            // = register j of lane 32 + c), so the determinant and its reciprocal run NEXT TO the broadcast of the two rows (rows89), not behind it; and the third product takes
            // F' (-1 / det) as one operand and adj(S) [F | f] as the other: behind the reciprocal the chain is one multiplication, then the MFMA (it was I = adj / det,
            // K = -(I F): three dependent operations).  K, kff and S^-1 for the other passes are formed off the chain.
#ifdef PG_F32
            const real bpb0 = rl(Cc[0], 40), S01 = rl(Cc[0], 41), bpb1 = rl(Cc[1], 41);
#else
            const real bpb0 = rl(Cc[2], 8), S01 = rl(Cc[2], 9), bpb1 = rl(Cc[2], 25);
#endif
            const real S00 = R0 + bpb0, S11 = R1 + bpb1;
            const real idet = frcp(S00 * S11 - S01 * S01), nid = -idet;      // (a reciprocal with ONE Newton step instead of two measured the same: 0.270 ms)
            real F0c, F1c; rows89(Cc, F0c, F1c);
            // [F | f]: column 10 carries f = rhat + Bbar'y, so ONE pair of products gives adj(S) F in columns 0..7 and adj(S) f in column 10
            const real Fx0 = fma(r0v, m10, F0c), Fx1 = fma(r1v, m10, F1c);
            const real X0 = S11 * Fx0 - S01 * Fx1, X1 = S00 * Fx1 - S01 * Fx0;
            // C += F' [K | kff]: the k = 2, 3 terms of the contraction vanish through the B operand (w0, w1 are zero there); lane c = 10 of the A operand feeds row 10, unread
            const real a3 = ((mg & 1) ? F1c : F0c) * nid;
            const real b3 = fma(X1, w1, X0 * w0);
            Cc = PG_MFMA(a3, b3, Cc);
            dKf[tKf * k] = ((mg & 1) ? X1 : X0) * nid;                              // K_k (columns 0..7), kff_k (column 10)
            dSi[tSi * k] = (lane == 0 ? S11 : (lane == 1 ? -S01 : S00)) * idet;     // S^-1: (I00, I01, I11)
            // (off the chain: what the multiplier of a pinned rate row is read from -- row 8 of C as it stands: F0, Bbar0'P Bbar0, S01, Bbar0'y in columns 0..10.  ONE store, and
            //  only in rounds that pin a stage: two stores with a three-way select and a lane read in every pass cost the interior point's matrix pass 15-19 %)
            if constexpr (EXR) { if (any_pin) *((mg == 0 && mc < 11) ? sF0 + 11 * k + mc : sDum + lane) = F0c; }
            real C0, C1; to_operands(Cc, C0, C1);
            V0 = C0 + add0; V1 = C1 + add1;
            // symmetrise P (see the note in the VALU version below) by a transpose through LDS -- every fourth stage: the A operand above reads P through its
            // own result, i.e. as P', so an antisymmetric rounding error e changes sign each stage and grows by |eig(Abar)|^2; four stages of that
            // are harmless for the N <= 32 horizons, and three LDS round trips in four leave the serial chain
            if (RING || (k & 3) == 0) {        // (long horizons -- the ring variant, N > 32 -- symmetrise every stage: at N = 50 the sparser schedule cost the interior point stragglers of 90 iterations)
                *(mc < 8 ? sP + 9 * mg + mc : sDum + lane) = V0;
                *(mc < 8 ? sP + 9 * (mg + 4) + mc : sDum + lane) = V1;
                wave_sync();
                const int tc = mc < 8 ? mc : 0;
                const real T0 = sP[9 * tc + mg], T1 = sP[9 * tc + mg + 4];
                V0 = mc < 8 ? real(0.5) * (V0 + T0) : V0; V1 = mc < 8 ? real(0.5) * (V1 + T1) : V1;
                wave_sync();
            }
        }
        if (nseg == 2 && seg == 0) { ck_V0 = V0; ck_V1 = V1; }
        }
        __syncthreads();
        if constexpr (CKPT) {      // (the feed-forward terms of the stages behind the checkpoint, as this pass left them: off the serial loop)
            if (!restart) { const int i = 2 * ck_k + lane; if (i < 2 * N) skf_ck[i] = skf[i]; }
        }
    };
#undef PG_MFMA
#else
    // Riccati matrix pass (once per IPM iteration): lane (li, lj) owns P[li][lj]; two LDS round trips per stage, branch-free body
    auto riccati_matrices = [&]() {
        real Pij = qbase[qmul * N];
        real pvec = sq[8 * N + r8];                  // predictor's backward vector recursion rides along (same stage order)
        const int ljb = lj < 2 ? lj : 0;
        ring_prime(N - 1, -1);
#pragma unroll 1
        for (int k = N - 1; k >= 0; k--) {
            ring_step(k, -1);
            const real* Ak = ring_slot(k); const real* Bk = Ak + SB_B; const real* ck = Ak + SB_C;
            sP[9 * li + lj] = Pij;
            // stage constants (independent of the recursion): issue their reads before the barrier
            real acol[6], arow[6], aug6[6], bk0[6], bk1[6];
#pragma unroll
            for (int m = 0; m < 6; m++) { acol[m] = Ak[SB_ROW * m + lj]; arow[m] = Ak[SB_ROW * m + li]; aug6[m] = lj == 2 ? ck[m] : Bk[2 * m + ljb]; bk0[m] = Bk[2 * m]; bk1[m] = Bk[2 * m + 1]; }
            real qh = qbase[qmul * k], R0 = sR[2 * k], R1 = sR[2 * k + 1];
            real a6v[6];
#pragma unroll
            for (int m = 0; m < 6; m++) a6v[m] = Ak[SB_ROW * m + r8];
            real qkv = sq[8 * k + r8], r0v = sr[2 * k], r1v = sr[2 * k + 1];
            wave_sync();
            real prow[8];
#pragma unroll
            for (int m = 0; m < 8; m++) prow[m] = sP[9 * li + m];
            real ma = lj == 6 ? prow[6] : (lj == 7 ? prow[7] : real(0.0));
            real aug = lj == 0 ? prow[6] : (lj == 1 ? prow[7] : real(0.0));
#pragma unroll
            for (int m = 0; m < 6; m++) { ma += prow[m] * acol[m]; aug += prow[m] * aug6[m]; }
            sMT[9 * lj + li] = ma;
            *(lj < 3 ? sMT + 9 * (8 + lj) + li : sDum + lane) = aug;
            *(lj == 2 ? sMc + 8 * k + li : sDum + lane) = aug;
            // S = Rhat + Bbar' (P Bbar): the two columns of P Bbar sit in the `aug` registers of lanes (m, 0) and (m, 1); broadcasting them through SGPRs
            // keeps 16 wave-uniform doubles out of LDS and takes the 2x2 solve off the path that waits for the exchange below
            real S00 = R0 + rl(aug, 48), S01 = rl(aug, 49), S11 = R1 + rl(aug, 57);
#pragma unroll
            for (int m = 0; m < 6; m++) { const real m0 = rl(aug, 8 * m), m1 = rl(aug, 8 * m + 1); S00 += bk0[m] * m0; S01 += bk0[m] * m1; S11 += bk1[m] * m1; }
            const real idet = frcp(S00 * S11 - S01 * S01);
            wave_sync();
            real cj[8], ci[8];
#pragma unroll
            for (int m = 0; m < 8; m++) { cj[m] = sMT[9 * lj + m]; ci[m] = sMT[9 * li + m]; }
            real Fj0 = cj[6], Fj1 = cj[7], Fi0 = ci[6], Fi1 = ci[7];
#pragma unroll
            for (int m = 0; m < 6; m++) { Fj0 += bk0[m] * cj[m]; Fj1 += bk1[m] * cj[m]; Fi0 += bk0[m] * ci[m]; Fi1 += bk1[m] * ci[m]; }
            real I00 = S11 * idet, I01 = -S01 * idet, I11 = S00 * idet;
            real K0 = -(I00 * Fj0 + I01 * Fj1), K1 = -(I01 * Fj0 + I11 * Fj1);
            *(li < 2 ? sK + 16 * k + 8 * li + lj : sDum + lane) = li == 0 ? K0 : K1;
            *(lane < 3 ? sSi + 4 * k + lane : sDum + lane) = lane == 0 ? I00 : (lane == 1 ? I01 : I11);
            real pn = qh + Fi0 * K0 + Fi1 * K1 + (li == 6 ? cj[6] : (li == 7 ? cj[7] : real(0.0)));
#pragma unroll
            for (int m = 0; m < 6; m++) pn += arow[m] * cj[m];
            // symmetrise (lane (j,i) holds the transposed entry): without it the antisymmetric rounding error of the recursion is amplified by
            // |eig(Abar)|^2 per stage and wrecks long horizons whose linearised dynamics are open-loop unstable (N = 50, saturated tires)
            Pij = real(0.5) * (pn + __shfl(pn, 8 * lj + li));      // (k == 0: never used)
            // ---- vector recursion of the predictor: y = Mc_k + p_{k+1} (Mc_k = column 10 of the augmented product, in sMT[90..97]) ----
            real yi = sMT[90 + r8] + pvec;
            real y[8];
#pragma unroll
            for (int m = 0; m < 8; m++) y[m] = rl(yi, m);
            real f0 = r0v + y[6], f1 = r1v + y[7], f0b = real(0.0), f1b = real(0.0);
            real acc = qkv + (r8 == 6 ? y[6] : (r8 == 7 ? y[7] : real(0.0))), accb = real(0.0);
#pragma unroll
            for (int m = 0; m < 6; m += 2) {
                f0 += bk0[m] * y[m]; f0b += bk0[m + 1] * y[m + 1]; f1 += bk1[m] * y[m]; f1b += bk1[m + 1] * y[m + 1];
                acc += a6v[m] * y[m]; accb += a6v[m + 1] * y[m + 1];
            }
            f0 += f0b; f1 += f1b;
            // K[c][r8] for this lane's component: K0/K1 above are K[c][lj] and lj == r8
            pvec = (acc + accb) + (K0 * f0 + K1 * f1);
            *(lane < 2 ? skf + 2 * k + lane : sDum + lane) = lane == 0 ? -(I00 * f0 + I01 * f1) : -(I01 * f0 + I11 * f1);
        }
        __syncthreads();
    };
#endif
    // (the lane-role split that pays in the roll-out -- gain rows in their own lanes -- was tried here too: fewer instructions, but the extra SGPR hop sits on the
    // serial chain y -> f -> p and the pass got 12 % slower; this pass is chain-bound, the roll-out was issue-bound)
    // Riccati vector pass backward: lane r8 holds p_{k+1}[r8]; p_k = qhat_k + Abar' y + K' f, y = Mc_k + p_{k+1}, f = rhat + Bbar' y, kff = -Sinv f
    auto riccati_vectors = [&](bool delta = false) {
        real pi = sq[8 * N + r8];
        ring_prime(N - 1, -1);
#pragma unroll 1
        for (int k = N - 1; k >= 0; k--) {
            ring_step(k, -1);
            const real* Ak = ring_slot(k); const real* Bk = Ak + SB_B;
            real a6[6], b0[6], b1[6];
#pragma unroll
            for (int m = 0; m < 6; m++) { a6[m] = Ak[SB_ROW * m + r8]; b0[m] = Bk[2 * m]; b1[m] = Bk[2 * m + 1]; }
            real mc = delta ? real(0.0) : sMc[8 * k + r8], qk = sq[8 * k + r8], k0 = sK[16 * k + r8], k1 = sK[16 * k + 8 + r8], r0 = sr[2 * k], r1 = sr[2 * k + 1];
            real I00 = sSi[4 * k], I01 = sSi[4 * k + 1], I11 = sSi[4 * k + 2];
            real yi = mc + pi;
            real y[8];
#pragma unroll
            for (int m = 0; m < 8; m++) y[m] = rl(yi, m);
            // (Bbar0'y is summed on its own and rhat0 added last: on a pinned stage rhat0 is BIGP x a rounding error and would swallow the sum the multiplier needs)
            real by0 = y[6], f1 = r1 + y[7], f0b = real(0.0), f1b = real(0.0);
            real acc = qk + (r8 == 6 ? y[6] : (r8 == 7 ? y[7] : real(0.0))), accb = real(0.0);
#pragma unroll
            for (int m = 0; m < 6; m += 2) {
                by0 += b0[m] * y[m]; f0b += b0[m + 1] * y[m + 1]; f1 += b1[m] * y[m]; f1b += b1[m + 1] * y[m + 1];
                acc += a6[m] * y[m]; accb += a6[m + 1] * y[m + 1];
            }
            by0 += f0b; f1 += f1b;
            const real f0 = r0 + by0;
            pi = (acc + accb) + (k0 * f0 + k1 * f1);          // (k == 0: never used)
            *(lane < 2 ? skf + 2 * k + lane : sDum + lane) = lane == 0 ? -(I00 * f0 + I01 * f1) : -(I01 * f0 + I11 * f1);
            if constexpr (EXR) { if (any_pin) *(lane == 2 ? sF0 + 11 * k + 10 : sDum + lane) = by0; }      // Bbar0'y of this pass
        }
        __syncthreads();
    };
    real xn[8], vn0, vn1, sn1, sn2, snh, last_dmax = real(0.0);
    auto newton_point = [&](real* tplus, bool delta = false) {
#pragma unroll
        for (int m = 0; m < 8; m++) xn[m] = (delta ? xn[m] : real(0.0)) + sx[8 * (s + 1) + m];
        vn0 = (delta ? vn0 : real(0.0)) + sv[2 * s]; vn1 = (delta ? vn1 : real(0.0)) + sv[2 * s + 1];
        if (__builtin_amdgcn_readfirstlane((int)delta)) {      // size of the correction (largest component of any stage): the refinement has converged when it is small
            real dm = real(0.0);
#pragma unroll
            for (int m = 0; m < 8; m++) dm = fmax(dm, act ? fabs(sx[8 * (s + 1) + m]) : real(0.0));
            last_dmax = wave_max(dm);
        }
        sn1 = -(e_c10 * xn[2] + e_c11 * xn[3] + e_g1) * e_d1;
        sn2 = -(e_c20 * xn[2] + e_c21 * xn[3] + e_g2) * e_d2;
        snh = wall_on ? -(e_ch0 * xn[5] + e_gh) * e_dh : (hji_on ? -(e_ch0 * xn[6] + e_ch1 * xn[7] + e_gh) * e_dh : real(0.0));
        slacks(xn, vn0, sn1, sn2, snh, tplus);
    };

    int it_total = 0, ptr_n = 0;
    // Warm start of the ACTIVE SET (the reference's warm start is OSQP's: previous (x, y) as the initial iterate, src/coupled_lat_long.jl:218 WarmStart = true).
    // An instance whose previous step ended in a solved QP first tries the polish directly from that step's active set and multipliers on the NEW QP data
    // (attempt -1): in closed loop the set rarely changes from one 10 ms step to the next, and a verified round IS the exact optimum of the new QP whatever the
    // guess was, so nothing is lost in accuracy; if the rounds do not verify, the interior point runs as for a cold instance.
    real* const Lst = O.lam + ((size_t)b * N + s) * NROW;
    bool warm = C.polish && C.warm_polish && prev_solved != 0 && prev_status == PG_SOLVED;
    // ... unless the previous set holds nothing but the pivots of the eliminated slacks: that IS the empty set, which attempt -1 starts from as well -- with the clipped
    // roll-out's guess should the new optimum have rows after all (closed loop of the headline batch, 40 steps: 6.38 -> see EXPERIMENTS.md 10.9)
    if (warm && C.clip_guess != 0 && C.warm_trivial_cold != 0) {
        const unsigned piv = (1u << 10) | (1u << 11) | (hji_on ? (1u << 15) : 0u) | (wall_on ? (1u << 2) : 0u);
        if (!__any(act && (((unsigned)O.active[(size_t)b * N + s]) & ~piv) != 0u)) warm = false;
    }
    // attempt -1: the polish from the EMPTY set (cold instances, and warm ones whose previous set did not verify).  Not where the safety row is violated at the
    // current control: its weight (W_HJI) then overrides the tracking cost, the optimum is close to bang-bang (rate rows of both signs, force bounds, soft rows all
    // change together) and the add / drop iteration turns over dozens of rows per round -- the interior point needs its usual 8 iterations there
    const bool hji_hot = C.has_hji && M0 * sx0[6] + M1 * sx0[7] + hji_b < real(0.0);
    // (fp32 with a violated safety row: the polish behind the interior point keeps every held row in the penalty form.  Its working sets are close to bang-bang -- rate rows of
    //  both signs next to force bounds -- and with the rate rows pinned 5 instead of 2 of the 4096 config-3 answers ended unverified, one of them 1.5e-2 off: single precision)
    exr = EXR && !(sizeof(real) == 4 && hji_hot);
    // hji_seed (option "hji_seed", experiment of round 4): a violated safety row does NOT send the instance to the interior point; its rounds start from a SEEDED working set
    // instead of the empty one -- the row held at its two stages with its slack free (an exact penalty: the multiplier of such a row IS the linear cost W_HJI of its slack)
    // and, with hji_seed = 2, the steering-rate row of those stages in the direction that relieves it
    const bool guess = C.polish && C.cold_guess > 0 && (!hji_hot || C.hji_seed > 0);
    int last_nchg = 0, good_steps = 0; real last_tmax = real(0.0); bool refine_only = false;
    int dbg_stage = -1, dbg_bit = -1, dbg_nadd = 0, dbg_ndrop = 0;      // diagnostic build: first stage / row that joined the set in the last check, rows added / dropped
    bool warm_attempt = false, from_prev = false;          // warm_attempt: a polish without an interior point in front (attempts -2, -1); from_prev: attempt -2
    for (int attempt = listm ? 0 : (warm ? -2 : (guess ? -1 : 0)); attempt < (IPM ? 2 : 0); attempt++) {
    if (attempt == -1 && !guess) continue;
    rp0 = real(0.0); phi = real(1.0); pmode = 0; pstat = 0; polish_gave_up = false; amask_1ago = 0xFFFFFFFFu; amask_2ago = 0xFFFFFFFFu; cycle_broken = false; settle_used = false; settle_left = 0; good_steps = 0;
    warm_attempt = attempt < 0; from_prev = attempt == -2;
    if (!IPM || attempt < 0) {
        amask = (act && from_prev) ? (unsigned)O.active[(size_t)b * N + s] : 0u; mask_ipm = amask;
#pragma unroll
        for (int j = 0; j < NROW; j++) { if constexpr (IPM) { R.t[j] = real(1.0); R.corr[j] = real(0.0); } R.lam[j] = (act && from_prev && ((amask >> j) & 1u)) ? Lst[j] : real(0.0); }
        if (attempt == -1 && hji_hot && hji_on) {
            // can the violation be removed at node 2 at all?  What M u can gain there: the steering at its rate limit and Fx up to its bound (Fx has no rate limit, only a
            // quadratic cost on its change).  If yes the row ends up MET with its slack at zero (row and sigma >= 0 both held, the steering rate limit of the first stage
            // usually with them: steering is the cheap actuator); if not, the row stays violated at the exact-penalty multiplier W_HJI, Fx sits on its bound and the
            // steering runs at its rate limit over both stages (the two patterns cover 370 of the 380 violated rows of config 3: tools/gpu_config3_probe.py PG_C3_DUMP=1)
            const real viol = -(M0 * sx0[6] + M1 * sx0[7] + hji_b);
            const real reach = fabs(M0) * rl(M0 > real(0.0) ? bb[12] : bb[13], 0) + (M1 > real(0.0) ? M1 * (rl(bb[5], 0) - sx0[7]) : -M1 * (sx0[7] + rl(bb[2], 0)));
            const bool removable = C.hji_seed >= 4 && viol < reach;
            const unsigned rate_bit = (M0 > real(0.0)) ? (1u << 12) : (1u << 13);
            if (removable) {
                amask |= (1u << 14) | (1u << 15); R.lam[14] = real(0.5) * wh; R.lam[15] = real(0.5) * wh;
                if (s == 0) amask |= rate_bit;
            } else {
                amask |= 1u << 14; R.lam[14] = wh;
                if (C.hji_seed >= 2) amask |= rate_bit;
                if (C.hji_seed >= 3) amask |= (M1 > real(0.0)) ? (1u << 5) : (1u << 2);      // ... and the force bound the row pushes Fx against
            }
            mask_ipm = amask;
        }
        pmode = 1; mu = real(0.0);
    } else if (attempt == 0) {
        if (!listm && O.n_todo && lane == 0) atomicAdd(O.n_todo, 1);      // (one-kernel launch: count the instances that need the interior point -- the host's choice between the split and the single launch)
        // ---- first attempt: v = 0 roll-out (dynamics- and rate-feasible), sigma just feasible; t = max(slack, tau); lambda = mu0 / t ----
        forward(std::false_type{});
        real xs[8];
#pragma unroll
        for (int m = 0; m < 8; m++) xs[m] = sx[8 * (s + 1) + m];
        real sl[NROW];
        slacks(xs, real(0.0), real(0.0), real(0.0), real(0.0), sl);
        // margin of the soft-row slacks above "just feasible": 0.1 for the coupled problem (tuned there: 6.6 iterations; 1 costs two more), 1 for the lateral one, whose
        // v = 0 roll-out drifts tens of metres off over 8 s -- the first Newton directions are that long, and a margin of 0.1 on the envelope / wall rows lets them
        // advance by 1e-2 of their length per iteration (25-30 iterations of tiny steps; with the wall rows some instances ran into the cap first: 110 iterations)
        const real sig0 = C.formulation == PG_DECOUPLED ? real(1.0) : real(0.1), tau = real(1e-4);
        real sg1 = fmax(real(0.0), -fmin(sl[6], sl[7])) + sig0, sg2 = fmax(real(0.0), -fmin(sl[8], sl[9])) + sig0, sgh = wall_on ? fmax(real(0.0), -fmin(sl[0], sl[1])) + sig0 : (hji_on ? fmax(real(0.0), -sl[14]) + sig0 : real(0.0));
        slacks(xs, real(0.0), sg1, sg2, sgh, sl);
        if (act) {
#pragma unroll
            for (int m = 0; m < 8; m++) SXs[m] = xs[m];
            SGs[0] = sg1; SGs[1] = sg2; SGs[2] = sgh;
        }
#pragma unroll
        for (int j = 0; j < NROW; j++) {
            bool on = act && j < nrows;
            real tj = on ? fmax(sl[j], tau) : real(1.0);
            R.t[j] = tj; R.lam[j] = on ? C.ipm_mu0 / tj : real(0.0); R.corr[j] = real(0.0);
            if (on) rp0 = fmax(rp0, tj - sl[j]);
        }
    } else {
        // ---- second attempt (only for instances the first start could not solve: long horizons whose linearised dynamics are open-loop
        // unstable make the v = 0 roll-out explode, |e| ~ 500 m at N = 50): least-squares start.  One Newton solve with every row replaced
        // by a unit-weight quadratic penalty (closed-loop roll-out, bounded), then a UNIFORM shift that makes every slack >= 1. ----
#pragma unroll
        for (int j = 0; j < NROW; j++) { R.t[j] = real(1.0); R.lam[j] = (act && j < nrows) ? real(1.0) : real(0.0); R.corr[j] = real(0.0); it_[j] = real(1.0); }
        assemble(real(0.0), true);
        __syncthreads();
        riccati_matrices();
        forward(std::true_type{});
        real tp[NROW];
        newton_point(tp);
        real tmin = PG_BIG;
#pragma unroll
        for (int j = 0; j < NROW; j++) tmin = fmin(tmin, (act && j < nrows) ? tp[j] : PG_BIG);
        tmin = wave_min(tmin);
        const real shift = tmin < real(1.0) ? real(1.0) - tmin : real(0.0);
        if (act) {
#pragma unroll
            for (int m = 0; m < 8; m++) SXs[m] = xn[m];
            SGs[0] = sn1; SGs[1] = sn2; SGs[2] = snh;
        }
#pragma unroll
        for (int j = 0; j < NROW; j++) {
            bool on = act && j < nrows;
            real tj = on ? tp[j] + shift : real(1.0);
            R.t[j] = tj; R.lam[j] = on ? C.ipm_mu0 / tj : real(0.0); R.corr[j] = real(0.0);
        }
        rp0 = shift;
    }
    stamp(5);
    rp0 = wave_max(rp0);
    status = warm_attempt ? PG_SOLVED : PG_MAX_ITER;
    const int iter_cap = attempt <= 0 ? C.ipm_max_iter : 3 * C.ipm_max_iter;
    const int round_cap = attempt == -1 ? C.cold_guess : PG_POLISH_ROUNDS;
    // a set that moves by a row or two per round on a nearly feasible point is a ramp being extended or released one stage at a time (the multiplier of the next row
    // only changes sign once the previous one has left): it gets there, and eight more 25 us rounds are far cheaper than the interior point they avoid
    // (a seeded attempt on a violated safety row -- hji_seed -- either verifies within a few working sets or not at all: its cap is hji_rounds, without the extension)
    auto over_cap = [&](int pass) { if (attempt == -1 && hji_hot && C.hji_rounds > 0) return pass > C.hji_rounds; return pass > round_cap + (((warm_attempt || sizeof(real) == 8) && last_tmax < real(1.0)) ? (last_nchg <= 2 ? (sizeof(real) == 8 ? 16 : 8) : (last_nchg <= PG_PROGRESS_ROWS ? 8 : 0)) : 0); };      // (also for the warm attempt: a fall-back to the cold start costs ten rounds, and the slowest instance sets the kernel time)
    // Active-set polish (OSQP-style, on the stage-structured problem).  The interior point approaches nearly degenerate rows (slack and multiplier both ~ sqrt(mu))
    // like sqrt(mu), so its iterate can sit 1e-6 away from the optimum at any tolerance fp64 rounding allows.  Once it has converged, the rows with
    // lambda > t are held as EQUALITIES (augmented Lagrangian with penalty rho, multiplier estimates y = lambda), every other row is dropped, and the
    // resulting equality-constrained LQ problem is solved by the SAME passes: one "predictor" (matrix pass) + one "corrector" (vector pass) per round, a
    // multiplier update  y <- y - rho t(z+)  after each.  A round is accepted when the point is primal feasible (inactive rows t >= -ptol, active rows
    // |t| <= ptol) and dual feasible (y >= 0); otherwise violated rows join the set, rows with y < 0 leave it, and the round repeats (PG_POLISH_ROUNDS at most).  A
    // polish that does not verify leaves the interior-point iterate in place.
    auto enter_polish = [&]() {
        amask = 0;
#pragma unroll
        for (int j = 0; j < NROW; j++) if (act && j < nrows && R.lam[j] > R.t[j]) amask |= (1u << j);
        mask_ipm = amask;
#pragma unroll
        for (int j = 0; j < NROW; j++) R.lam[j] = ((amask >> j) & 1u) ? R.lam[j] : real(0.0);
        pmode = 1;
    };
    // after a polish solve (tp = slacks at the new point): multiplier update of the active rows, then the verification.  Returns 0 = verified (solution
    // stored, pstat set), 1 = same set but the active rows are not yet at t = 0 within `ttol` (refine), 2 = the active set changed, 3 = the sets cycle.
    auto polish_check = [&](const real* tp, real ttol, real dmax = real(0.0), bool delta = false) -> int {
        unsigned add = 0, drop = 0; bool settled = !(dmax > dtol);
        // (a roll-out that clipped the steering rate is not the optimum of its working set: no multiplier update, no drops -- its clipped transitions join the set)
        const bool clipped = __any(clip_mask != 0u);
        if (clipped) { add = act ? (clip_mask & ~amask) : 0u; settled = false; clip_used = true; }
        if (exr && !clipped) {      // pinned rate row: the multiplier IS minus / plus the gradient of the Lagrangian in the pinned input at the roll-out's point (t = 0 exactly)
            const bool h12 = act && (amask & (1u << 12)) != 0u, h13 = act && !h12 && (amask & (1u << 13)) != 0u;
            if (__any(h12 || h13)) {
                const real* F = sF0 + 11 * s; const real* xs = sx + 8 * s; const real* ax = F + 8;
                real g = ax[2], g2 = real(0.0);
#pragma unroll
                for (int m = 0; m < 8; m += 2) { g += F[m] * xs[m]; g2 += F[m + 1] * xs[m + 1]; }
                const real v0s = sv[2 * s], v1s = sv[2 * s + 1], vpin = h12 ? bb[12] : -bb[13];
                g = (g + g2) + ((delta ? Rd0 * vpin : (Rd0 + ax[0]) * v0s) + ax[1] * v1s);
                R.lam[12] = h12 ? -g : R.lam[12]; R.lam[13] = h13 ? g : R.lam[13];      // (selects: a conditional store to either slot sends both to scratch)
            }
        }
#pragma unroll
        for (int j = 0; j < NROW; j++) {
            const bool on = act && j < nrows, a = (amask >> j) & 1u;
            if (a && !clipped && !(exr && (j == 12 || j == 13))) R.lam[j] -= rho * tp[j];
            if (on && a && !clipped && R.lam[j] < real(0.0)) drop |= 1u << j;
            if (on && a && !(fabs(tp[j]) <= ttol)) settled = false;            // written so that a NaN never verifies
            if (on && !a && !(tp[j] >= -ptol)) add |= 1u << j;
        }
        const bool changed = __any((add | drop) != 0u), conv = __all(settled);
        {   // rows that cannot be active (addable) or cannot be active next to a row that is, or is about to be, in the set (excl)
            unsigned addf = add & addable;
            const unsigned wa = amask | addf, wa_p = __shfl_up(wa, 1);
            if ((wa & (1u << 3)) && (excl & 1u)) addf &= ~(1u << 13);
            if ((wa & (1u << 4)) && (excl & 2u)) addf &= ~(1u << 12);
            if ((wa & (1u << 3)) && (wa_p & (1u << 3)) && (excl & 4u)) addf &= ~(3u << 12);
            if ((wa & (1u << 4)) && (wa_p & (1u << 4)) && (excl & 8u)) addf &= ~(3u << 12);
            // at most one NEW row per slack group and round, the more violated one: two rows that share a slack, both held with the slack free, pin a combination of
            // the states hard -- a pair picked up together at an overshooting point has sent the multipliers of a whole (correct) ramp of rate rows negative
            if ((addf & 0x00C0u) == 0x00C0u) addf &= ~(tp[6] <= tp[7] ? (1u << 7) : (1u << 6));
            if ((addf & 0x0300u) == 0x0300u) addf &= ~(tp[8] <= tp[9] ? (1u << 9) : (1u << 8));
            if (wall_on && (addf & 0x0003u) == 0x0003u) addf &= ~(tp[0] <= tp[1] ? 2u : 1u);
            add = addf;
        }
        if constexpr (PROF) {
            const unsigned long long bl = __ballot(add != 0u);
            dbg_stage = bl ? __ffsll((long long)bl) - 1 : -1;
            const unsigned a0 = __shfl(add, dbg_stage < 0 ? 0 : dbg_stage);
            dbg_bit = a0 ? __ffs((int)a0) - 1 : -1;
            dbg_nadd = (int)wave_sum(real(__popc(add))); dbg_ndrop = (int)wave_sum(real(__popc(drop)));
        }
        const bool stalled = changed && !__any((add | drop) != 0u);      // only rows that cannot be active are violated (then another row is too: not expected)
        {   // progress of the set iteration, for the round cap and the bail-out below
            real tmx = real(0.0);
#pragma unroll
            for (int j = 0; j < NROW; j++) tmx = fmax(tmx, (act && ((amask >> j) & 1u)) ? fabs(tp[j]) : real(0.0));
            last_tmax = wave_max(tmx);
            last_nchg = (int)wave_sum(real(__popc(add | drop)));
        }
        // settling (see the cycle rule below): the working set stays as it is until its held rows are at t = 0, i.e. until the multiplier estimates have converged
        if (settle_left > 0) { if (!conv) { settle_left--; return 1; } settle_left = 0; }
        if (!changed && conv) {
            if (act) {
#pragma unroll
                for (int m = 0; m < 8; m++) SXs[m] = xn[m];
                SGs[0] = sn1; SGs[1] = sn2; SGs[2] = snh;
            }
            pstat = pmode;
            return 0;
        }
        if (!changed) return 1;
        // a working set whose rows cannot be met together (|t| of order one on rows held as equalities), or one that turns over dozens of rows after the first round
        // (healthy iterations add ~12 rows in round 1 and a handful later; 25+ is an iteration that has lost the plot), is not worth iterating on when there is no
        // interior-point iterate to protect: a polish without an interior point in front moves on to its fall-back at once
        if (warm_attempt && (!(last_tmax < real(1.0)) || (pmode >= 2 && last_nchg > (2 * N) / 3))) return 3;
        // a violated SOFT row joins the set with its slack FREE (the sigma >= 0 row of its group leaves): held together with sigma = 0 it would be a hard equality,
        // inconsistent wherever the state cannot move (the envelope rows of the first stages, whose states the current control already fixes); if the slack
        // comes out negative, its sigma >= 0 row is violated and comes back next round
        if (attempt == -1) {
        if (add & 0x00C0u) drop |= amask & (1u << 10);
        if (add & 0x0300u) drop |= amask & (1u << 11);
        if (hji_on && (add & (1u << 14))) drop |= amask & (1u << 15);
        if (wall_on && (add & 0x0003u)) drop |= amask & (1u << 2);
        }
        unsigned next = (amask & ~drop) | add;
        // a set that comes back after two rounds (A -> B -> C -> A) is a cycle between inconsistent guesses (degenerate rows; typical of the weakly determined far
        // end of the N = 50 lateral horizon): further rounds would only repeat it.
        // One kind of A -> B -> A alternation has a known way out.  A soft row sits a hair inside its bound at the optimum (slack 1e-8 .. 1e-6: the interior point cannot tell, and hands it
        // over as active together with its sigma >= 0 row).  Held with sigma = 0 the pair is a hard equality whose sigma row gets a negative multiplier -> the sigma row
        // leaves; with sigma free the slack comes out (slightly) negative -> the sigma row returns; and so on.  The consistent third choice is the one neither set
        // tries: the soft row OUT, its sigma row in.  Taken once per polish.
        // (A -> B -> A alone is not given up on: the multipliers are still settling, and revisited sets often verify; A -> B -> C -> A is.)
        const bool cycle = __all(next == amask_2ago);
        if (!cycle && !cycle_broken && __all(next == amask_1ago)) {
            const unsigned tog = amask ^ next;
            unsigned nx = next;
            if (tog & (1u << 10)) nx = (nx | (1u << 10)) & ~0x00C0u;
            if (tog & (1u << 11)) nx = (nx | (1u << 11)) & ~0x0300u;
            if (hji_on && (tog & (1u << 15))) nx = (nx | (1u << 15)) & ~(1u << 14);
            if (wall_on && (tog & (1u << 2))) nx = (nx | (1u << 2)) & ~0x0003u;
            if (__any(nx != next)) { cycle_broken = true; next = nx; }
        }
        // A -> B -> C -> A with the held rows still far from t = 0 (fp32 at rho = 1e3: |t| ~ 1e-3 against polish_tol 1e-4) is a cycle of DECISIONS TAKEN ON UNCONVERGED
        // MULTIPLIERS -- one or two weakly active rows toggle on the sign of an estimate that is still moving.  Once per polish the set is kept for up to
        // PG_SETTLE_PASSES refinement passes (vector pass + correction roll-out each) and the decisions are taken again from settled multipliers; the alternative is the
        // interior point (7 iterations of two passes each, which for the ~7 such instances of a cold fp32 batch of 4096 was the duration of the whole launch).
        if ((cycle || stalled) && !settle_used && !(last_tmax <= ttol)) {      // (stalled: the only violated rows are ones the set cannot take -- the same unsettled estimates)
            settle_used = true; settle_left = PG_SETTLE_PASSES; amask_1ago = 0xFFFFFFFFu; amask_2ago = 0xFFFFFFFFu;
            return 1;
        }
        amask_2ago = amask_1ago; amask_1ago = amask;
        if (cycle || stalled) return 3;
        amask = next;
#pragma unroll
        for (int j = 0; j < NROW; j++) R.lam[j] = ((amask >> j) & 1u) ? R.lam[j] : real(0.0);
        return 2;
    };
    // with the polish on, the interior point only has to get close enough for the active set to show (polish_ipm_tol); if the polish does not verify from
    // there, the interior point resumes from the centred point (t, mu / t) and runs down to ipm_tol before the polish gets its second and last chance
    real tol_cur = (C.polish && C.polish_ipm_tol > C.ipm_tol) ? C.polish_ipm_tol : C.ipm_tol;
    auto polish_failed = [&]() -> bool {          // true: give up (keep the interior-point iterate); false: the interior point resumes
        if (!IPM || warm_attempt) { pstat = 0; status = PG_MAX_ITER; return true; }      // warm guess did not verify: on to the cold start
        pstat = -1;
        if (!(tol_cur > C.ipm_tol)) { polish_gave_up = true; return true; }
        tol_cur = C.ipm_tol; pmode = 0; status = PG_MAX_ITER; amask_1ago = 0xFFFFFFFFu; amask_2ago = 0xFFFFFFFFu; cycle_broken = false; settle_used = false; settle_left = 0;
#pragma unroll
        for (int j = 0; j < NROW; j++) R.lam[j] = (act && j < nrows) ? mu * frcp(R.t[j]) : real(0.0);
        return false;
    };
    // diagnostic build: one trace record per polish check of PG_DEBUG_INSTANCE behind the interior-point records (entries 128..255 of the trace region):
    // (attempt * 100 + pass number, outcome of the check, rows set in the working set, max |t| over its rows)
    auto ptrace = [&](int pc, const real* tp) {
        if constexpr (PROF) {
            if (b != C.dbg_instance) return;
            real mx = real(0.0), cnt = real(0.0);
#pragma unroll
            for (int j = 0; j < NROW; j++) if (act && ((amask >> j) & 1u)) { mx = fmax(mx, fabs(tp[j])); cnt += real(1.0); }
            mx = wave_max(mx); cnt = wave_sum(cnt);
            if (lane == 0 && ptr_n < 128) { real* tr = reinterpret_cast<real*>(prof + (size_t)B * 6) + 4 * (128 + ptr_n); tr[0] = real(100 * (attempt + 2) + pmode); tr[1] = real(pc + 100 * dbg_nadd + 10000 * dbg_ndrop); tr[2] = cnt + real(1000 * (dbg_stage + 1) + 100000 * (dbg_bit + 1)); tr[3] = mx; }
            ptr_n++;
        }
    };
    it = 0;
    while (true) {
        if (IPM && !pmode) {
            // (an attempt that is converging at its cap -- three steps in a row longer than one half -- gets twenty more iterations (caps of 20 and more: not the test settings that force the second start): the alternative is a second start
            // from scratch; long lateral horizons with wall rows spend 25-30 iterations on tiny steps before the iteration takes off)
            if (it >= iter_cap && !(iter_cap >= 20 && good_steps >= 3 && it < iter_cap + 20)) break;
            real musum = real(0.0);
#pragma unroll
            for (int j = 0; j < NROW; j++) musum += (act && j < nrows) ? R.t[j] * R.lam[j] : real(0.0);
            mu = wave_sum(musum) / ntot;
            if (!(mu == mu) || fabs(mu) > PG_BIG) { status = PG_NUMERICAL; break; }
            if (mu <= tol_cur && phi * fmax(rp0, real(1.0)) <= tol_cur) {
                status = PG_SOLVED;
                if (!C.polish || polish_gave_up) break;
                enter_polish();
            } else {
#pragma unroll
                for (int j = 0; j < NROW; j++) it_[j] = frcp(R.t[j]);
            }
        }
        real tp[NROW];
        real sg = real(0.0);
        if (!(pmode && refine_only)) {
        if (pmode) {
            // every stage-locally eliminated slack needs a pivot: a group without an active row gets its sigma >= 0 row (the linear cost pushes sigma down to it)
            // (a pivot that is the only active row of its group has a KNOWN multiplier, the linear cost coefficient of its slack: starting the augmented Lagrangian
            // there instead of at zero spares the refinement pass whose only job would be to find it -- every instance served from the empty set had one)
            if (!(amask & 0x04C0u)) { amask |= 1u << 10; if (act) R.lam[10] = wb; }
            if (!(amask & 0x0B00u)) { amask |= 1u << 11; if (act) R.lam[11] = wr; }
            if (hji_on && !(amask & 0xC000u)) { amask |= 1u << 15; R.lam[15] = wh; }
            if (wall_on && !(amask & 0x0007u)) { amask |= 1u << 2; R.lam[2] = ww; }
            // rate rows that would carry an unbroken run from the first transition past the steering bound (see `overshoot`)
            const unsigned long long run_up = __ballot(act && ((amask >> 12) & 1u)), run_dn = __ballot(act && ((amask >> 13) & 1u));
            const int end_up = __ffsll((long long)~run_up) - 1, end_dn = __ffsll((long long)~run_dn) - 1;      // first stage outside the run (lanes >= N are never in it)
            if (s < end_up) amask &= ~(overshoot & (1u << 12));
            if (s < end_dn) amask &= ~(overshoot & (1u << 13));
        }

        // ---- predictor (sigma = 0, no correction) / first polish solve ----
#pragma unroll
        for (int j = 0; j < NROW; j++) R.corr[j] = real(0.0);
        stamp(0);
        tl_mark(0);
        assemble(real(0.0), true);
        __syncthreads();
        stamp(1);
        tl_mark(1);
        {
            bool restart = false;
            if constexpr (CKPT) {
                if (C.ck_riccati != 0) {
                    const unsigned piv = (1u << 10) | (1u << 11) | (hji_on ? (1u << 15) : 0u) | (wall_on ? (1u << 2) : 0u);
                    const unsigned long long held = __ballot(act && (amask & ~piv) != 0u);
                    const int top = held ? 63 - __builtin_clzll(held) : -1;
                    restart = top < ck_k - 1 && ck_top < ck_k - 1;      // (the rows of stage lane s sit on node s + 1: P_{ck_k} holds those of lane ck_k - 1)
                    if (!restart) ck_top = top;
                }
            }
            ck_restart = restart;
            riccati_matrices();                // matrix recursion + the predictor's vector recursion
            ck_restart = false;
        }
        stamp(2);
        tl_mark(2);
        clip_mask = 0u; clip_now = C.clip_guess != 0 && !clip_off && attempt == -1 && pmode == 1 && !hji_hot;
        forward(std::true_type{});
        clip_now = false;
        stamp(4);
        tl_mark(3);
        newton_point(tp);
        if (IPM && !pmode) {
            // step to the boundary: alpha_max = 1 / max_j( -dt_j / t_j, -dl_j / lam_j )  (only rows that move towards the boundary are positive)
            real rmax = real(0.0);
#pragma unroll
            for (int j = 0; j < NROW; j++) {
                bool on = act && j < nrows;
                real dt_ = tp[j] - R.t[j], dl_ = -(R.lam[j] * it_[j]) * tp[j];     // lambda+ - lambda with sigma*mu = 0, corr = 0
                R.corr[j] = dt_ * dl_;
                real rj = fmax(-dt_ * it_[j], -dl_ * frcp(R.lam[j]));
                rmax = fmax(rmax, on ? rj : real(0.0));
            }
            rmax = wave_max(rmax);
            real aaff = rmax > real(1.0) ? real(1.0) / rmax : real(1.0);
            // rounding floor: once mu is within 1e4x of the tolerance and the affine direction can no longer move (step to the boundary < 0.3),
            // further iterations only add noise (observed on long, ill-conditioned horizons): accept the iterate as it stands
            if (mu <= real(1e4) * C.ipm_tol && aaff < real(0.3) && phi * fmax(rp0, real(1.0)) <= C.ipm_tol) {
                status = PG_SOLVED;
                if (!C.polish || polish_gave_up) break;
                enter_polish();
                continue;
            }
            real msum = real(0.0);
#pragma unroll
            for (int j = 0; j < NROW; j++) {
                real dt_ = tp[j] - R.t[j], dl_ = -(R.lam[j] * it_[j]) * tp[j];
                msum += (act && j < nrows) ? (R.t[j] + aaff * dt_) * (R.lam[j] + aaff * dl_) : real(0.0);
            }
            real mu_aff = wave_sum(msum) / ntot;
            sg = fmin(mu_aff / mu, real(1.0)); sg = sg * sg * sg;          // Mehrotra centring parameter, never above 1
            if (PROF && b == C.dbg_instance && lane == 0 && it_total + it < 256) { real* tr = reinterpret_cast<real*>(prof + (size_t)B * 6) + 4 * (it_total + it); tr[1] = aaff; }
        } else {
            // first polish solve done: with multiplier estimates as good as the interior point's, it usually verifies at once (no refinement needed)
            const int pc = polish_check(tp, real(0.01) * ptol);
            tl_mark(4);
            ptrace(pc, tp);
            if (pc == 0) break;
            if (pc == 3) { if (polish_failed()) break; continue; }
            if (pc == 2) { if (over_cap(++pmode) && polish_failed()) break; continue; }      // the set changed: next round directly
        }
        }
        refine_only = false;
        // ---- corrector / polish refinement ----
        assemble(sg * mu, false);
        // Polish: the corrector is solved as a CORRECTION to the point the predictor returned (x = xbar + d): same matrices, linear terms = the gradient of the
        // stage costs AT xbar with the updated multipliers (qhat + Qhat xbar, rhat + Rhat vbar), no affine dynamics term (the roll-out that produced xbar satisfies
        // the dynamics to rounding), d_0 = 0.  In exact arithmetic this is the same point as the absolute solve; in floating point it is one step of iterative
        // refinement for free -- the Riccati recursion with the penalty rho on the working set carries entries of size rho, and where weakly curved directions
        // couple with penalised ones its absolute solution is only good to ~ eps rho / curvature (a recorded QP of the stress regime: controls 4e-3 off at rho = 1e7,
        // tests/test_gpu_qp_replay.py); the correction is computed from a residual evaluated at the actual point, so that error is squared.
        const bool dlt = pmode != 0;
        if (dlt && act) {
            real* qo = sq + 8 * (s + 1); const real* Qo = sQ + 10 * (s + 1);
            qo[0] = Qo[0] * xn[0]; qo[1] += Qo[1] * xn[1]; qo[2] += Qo[2] * xn[2] + Qo[8] * xn[3]; qo[3] += Qo[3] * xn[3] + Qo[8] * xn[2];
            qo[4] = Qo[4] * xn[4]; qo[5] = (wall_on ? qo[5] : real(0.0)) + Qo[5] * xn[5];
            qo[6] += Qo[6] * xn[6] + Qo[9] * xn[7]; qo[7] += Qo[7] * xn[7] + Qo[9] * xn[6];
            sr[2 * s] += sR[2 * s] * vn0; sr[2 * s + 1] = sR[2 * s + 1] * vn1;
        }
        __syncthreads();
        stamp(1);
        riccati_vectors(dlt);
        if (dlt && act) {      // (the entries assemble() never rewrites go back to their constants for the next absolute solve)
            real* qo = sq + 8 * (s + 1);
            qo[0] = real(0.0); qo[4] = real(0.0); if (!wall_on) qo[5] = real(0.0);
            sr[2 * s + 1] = real(0.0);
        }
        stamp(3);
        forward(std::true_type{}, dlt);
        stamp(4);
        newton_point(tp, dlt);
        if (IPM && !pmode) {
            real rmax = real(0.0);
#pragma unroll
            for (int j = 0; j < NROW; j++) {
                bool on = act && j < nrows;
                real dt_ = tp[j] - R.t[j], dl_ = (sg * mu - R.corr[j]) * it_[j] - (R.lam[j] * it_[j]) * tp[j];
                tp[j] = dl_;                              // keep d(lambda); d(t) is recomputed from the stage point below
                R.corr[j] = dt_;
                real rj = fmax(-dt_ * it_[j], -dl_ * frcp(R.lam[j]));
                rmax = fmax(rmax, on ? rj : real(0.0));
            }
            rmax = wave_max(rmax);
            real alpha = rmax > real(0.995) ? real(0.995) / rmax : real(1.0);
            // rounding floor, third form: in exact arithmetic a step takes mu to (1 - alpha (1 - sigma)) mu; a step that would MULTIPLY mu near the tolerance is a Newton
            // direction computed at a conditioning fp64 no longer carries (W = lambda / t ~ 1e10) -- the iterate at hand is as good as it gets, taking the step throws
            // it away (an N = 50 lateral instance: mu 1.3e-8 -> 3e-7 -> 3, attempt failed at its cap, 80 more iterations in the second: the whole batch waited 8 ms)
            if (!C.polish && mu <= (sizeof(real) == 8 ? real(1e5) : real(1e2)) * C.ipm_tol && phi * fmax(rp0, real(1.0)) <= C.ipm_tol) {
                real mnew = real(0.0);
#pragma unroll
                for (int j = 0; j < NROW; j++) mnew += (act && j < nrows) ? (R.t[j] + alpha * R.corr[j]) * (R.lam[j] + alpha * tp[j]) : real(0.0);
                mnew = wave_sum(mnew) / ntot;
                if (!(mnew <= real(4.0) * mu)) { status = PG_SOLVED; break; }      // (only with the polish off: with it on, the hand-over happens long before, and an
                                                                                   // instance whose polish gave up should get as close as the interior point can take it)
            }
#pragma unroll
            for (int j = 0; j < NROW; j++) {
                bool on = act && j < nrows;
                R.t[j] += on ? alpha * R.corr[j] : real(0.0); R.lam[j] += on ? alpha * tp[j] : real(0.0);
            }
            if (act) {
#pragma unroll
                for (int m = 0; m < 8; m++) { real c = SXs[m]; SXs[m] = c + alpha * (xn[m] - c); }
                real c1 = SGs[0], c2 = SGs[1], c3 = SGs[2];
                SGs[0] = c1 + alpha * (sn1 - c1); SGs[1] = c2 + alpha * (sn2 - c2); SGs[2] = c3 + alpha * (snh - c3);
            }
            phi *= (real(1.0) - alpha);
            good_steps = alpha > real(0.5) ? good_steps + 1 : 0;
            if (PROF && b == C.dbg_instance && lane == 0 && it_total + it < 256) {      // trace region behind the [B][6] cycle counters: (mu, aaff, sigma, alpha) per iteration
                real* tr = reinterpret_cast<real*>(prof + (size_t)B * 6) + 4 * (it_total + it);
                tr[0] = mu; tr[2] = sg; tr[3] = alpha;
            }
            if (mu > real(1e8) * C.ipm_mu0) break;          // diverging: give up on this start
            it++;
        } else {
            const int pc = polish_check(tp, ptol, last_dmax, true);
            ptrace(pc + 10, tp);
            if (pc == 0) break;
            // same set, but the rows are not at t = 0 yet or the correction was not small: another refinement pass on the same matrices (no predictor)
            refine_only = pc == 1;
            if ((pc == 3 || over_cap(++pmode)) && polish_failed()) { refine_only = false; break; }      // (set changed: another round)
        }
    }
    it_total += it;
    if (status == PG_SOLVED || status == PG_NUMERICAL) break;
    // a cold attempt that started from the clipped roll-out's set and did not verify is repeated once from the plain empty set (the rounds of rounds 2-3: one fp32
    // instance of 4096 needs it -- and the launch lasts as long as an instance that falls through to the interior point)
    if (attempt == -1 && clip_used && !clip_off) { clip_off = true; attempt = -2; }
    }   // attempts
    it = it_total;
    stamp(0);
    tl_mark(5);
    if (PG_TL && prof && lane == 0) {      // (the timeline also from the product's kernel in a -DPG_TIMELINE build: pg_debug_solve_cycles with the option "diag_timeline")
        if constexpr (PROF) for (int i = 0; i < 6; i++) prof[(size_t)b * 6 + i] = pc[i];
        // timeline record behind the trace region: wall clock (100 MHz) at entry and here, and where the wavefront ran (HW_ID | XCC_ID << 32)
        unsigned long long* tl = prof + (size_t)B * 6 + 1024 + (size_t)b * 3;
        tl[0] = t_entry; tl[1] = wall_clock64();
        tl[2] = (unsigned long long)__builtin_amdgcn_s_getreg((4 /*HW_ID*/) | (0 << 6) | (31 << 11)) | ((unsigned long long)__builtin_amdgcn_s_getreg((20 /*XCC_ID*/) | (0 << 6) | (31 << 11)) << 32);
    }
    if constexpr (!IPM) {
        if (status != PG_SOLVED) {      // not served by the rounds: left for the full kernel (nothing of this instance's outputs or warm-start state has been touched)
            if (lane == 0) O.todo[atomicAdd(O.n_todo, 1)] = b;
            return;
        }
    }
    if (status == PG_SOLVED && C.polish && pstat < 0) status = PG_SOLVED_UNVERIFIED;      // an interior-point iterate no active-set round could verify (pigeon_mpc.h)
    if (status == PG_SOLVED || status == PG_SOLVED_UNVERIFIED) {
        real Ux0 = sx0[1], Fx0 = sx0[7];
        if (Ux0 < C.cp.V_min || Ux0 > C.cp.V_max || Fx0 < C.fxmin_n) status = PG_INFEASIBLE_X0;
    }
    // ---- outputs ----
    real* SX = O.sol_x + (size_t)b * NN * 8;
    if (lane < 8) SX[lane] = sx0[lane];
    if (act) {
        unsigned mask = 0;
        if constexpr (IPM) {
#pragma unroll
            for (int j = 0; j < NROW; j++) if (j < nrows && R.lam[j] > R.t[j]) mask |= (1u << j);
        }
        if (pstat > 0) mask = amask;                               // the polish's verified set
        else if (pmode) mask = mask_ipm;                           // polish ended unverified with the multipliers overwritten: the interior point's set at hand-over
        O.active[(size_t)b * N + s] = (uint16_t)mask;
#pragma unroll
        for (int j = 0; j < NROW; j++) Lst[j] = R.lam[j];          // multipliers for the next step's warm polish
    }
    if (lane == 0) {
        // get_next_control: coupled_lat_long.jl:370-374 (node 2 of the reference = stage lane 0's node)
        real d = SXs[6] * C.un0, Fx = SXs[7] * C.un1;         // lane 0 is stage 0: SXs is node 2 of the reference
        if (C.formulation == PG_DECOUPLED) Fx = nodes[((size_t)b * NN + 1) * 10 + 7];      // decoupled_lat_long.jl:275-278: Fx of the seeded node 2
        real* U = O.u_out + (size_t)b * 3;
        U[0] = d; U[1] = Fx > real(0.0) ? Fx * C.veh.fwd_frac : Fx * C.veh.fwb_frac; U[2] = Fx > real(0.0) ? Fx * C.veh.rwd_frac : Fx * C.veh.rwb_frac;
        if (O.u_out2) { real* U2 = O.u_out2 + (size_t)b * 3; U2[0] = U[0]; U2[1] = U[1]; U2[2] = U[2]; }
        O.status[b] = status; O.iters[b] = it; O.mu[b] = mu; O.polish[b] = pstat;
        O.solved[b] = 1;      // model_predictive_control.jl:76: solved = true
    }
}

#include "pg_solve_lat.hip"         // the lateral formulation's own solve kernel (5-state stage, sixteen lanes per instance)

#ifdef PG_EXPERIMENTAL_SOLVE4      // four instances per wavefront: a measured negative result (EXPERIMENTS.md 4.1), kept out of the shipped libraries
#include "experimental/pg_solve4.hip"
#endif

}  // namespace pg
