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
#include "pg_device.hpp"

namespace pg {

struct DevCfg {
    pg_vehicle veh;
    pg_control_params cp;
    int Ns, Nl, N, NN;            // NN = N + 1 nodes
    double dt_short, dt_long;
    int use_correction_step, nsub;
    int alias_prev_ts;            // the reference's MPCTimeSteps passes `ts` as prev_ts too (model_predictive_control.jl:15): same array
    int has_hji;
    double hji_eps;
    double un0, un1;              // u_normalization (coupled_lat_long.jl:199)
    double fxmin_n;               // Fx_min / un1
    int qp_len;
    int ipm_max_iter;
    double ipm_tol, ipm_mu0;
    TrajView traj;
};

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
__global__ void k_time_steps(DevCfg C, int B, const double* __restrict__ t0, double* __restrict__ ts, double* __restrict__ dt, double* __restrict__ prev_ts) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double* T = ts + (size_t)b * C.NN; double* PT = prev_ts + (size_t)b * C.NN; double* D = dt + (size_t)b * C.N;
    for (int i = 0; i < C.NN; i++) PT[i] = T[i];                                   // :20
    double t = t0[b];
    double t0_long = t + C.Ns * C.dt_short;                                        // :21
    if (C.use_correction_step) t0_long = C.dt_long * ceil((t0_long + C.dt_short) / C.dt_long - 1.0);   // :23
    for (int i = 0; i <= C.Ns; i++) T[i] = t + C.dt_short * i;                     // :25
    for (int i = 1; i <= C.Nl; i++) T[C.Ns + i] = t0_long + C.dt_long * i;         // :26
    for (int i = 0; i < C.N; i++) D[i] = T[i + 1] - T[i];                          // :27-29
    if (C.alias_prev_ts) for (int i = 0; i < C.NN; i++) PT[i] = T[i];              // prev_ts IS ts in the reference (:15)
}

// ------------------------------------------------------------------------------------------------------------------
// math.jl:4-9
PG_DEV double seg_dist2(double ax, double ay, double bx, double by, double x, double y) {
    double vx = bx - ax, vy = by - ay;
    double lam = (vx * (x - ax) + vy * (y - ay)) / (vx * vx + vy * vy);
    lam = lam < 0.0 ? 0.0 : (lam > 1.0 ? 1.0 : lam);
    double px = (1.0 - lam) * ax + lam * bx, py = (1.0 - lam) * ay + lam * by;
    return (px - x) * (px - x) + (py - y) * (py - y);
}
// one wave per instance; strict '<' with lowest index winning ties == the reference's sequential scan (:71-79)
__global__ __launch_bounds__(256) void k_project(DevCfg C, int B, const double* __restrict__ state, double* __restrict__ sep) {
    int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= B) return;
    const TrajView& T = C.traj;
    double x = state[(size_t)wave * 6 + 0], y = state[(size_t)wave * 6 + 1];
    double best = INFINITY; int bi = 0x7fffffff;
    for (int i = lane; i < T.L - 1; i += 64) {
        double d2 = seg_dist2(T.E[i], T.N[i], T.E[i + 1], T.N[i + 1], x, y);
        if (d2 < best) { best = d2; bi = i; }           // i increases within a lane, so strict '<' keeps the lowest index
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        double ob = __shfl_xor(best, off); int oi = __shfl_xor(bi, off);
        if (ob < best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) {
        int i = bi;
        double vx = T.E[i + 1] - T.E[i], vy = T.N[i + 1] - T.N[i], wx = x - T.E[i], wy = y - T.N[i];
        double ds = sqrt(wx * wx + wy * wy - best);                                // :82
        double cr = vx * wy - vy * wx;
        double Ai = (T.V[i + 1] - T.V[i]) / (T.t[i + 1] - T.t[i]);
        double dt = fabs(Ai) < 1e-3 ? ds / T.V[i] : (sqrt(2.0 * Ai * ds + T.V[i] * T.V[i]) - T.V[i]) / Ai;
        double* o = sep + (size_t)wave * 4;
        o[0] = T.s[i] + ds; o[1] = sqrt(best) * sgn(cr); o[2] = T.t[i] + dt; o[3] = (double)i;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// nodes record per node: q[6], u[2] (physical units), V, kappa  -> 10 doubles
__global__ void k_nodes(DevCfg C, int B, const double* __restrict__ state, const double* __restrict__ control, const double* __restrict__ toff,
                        const int* __restrict__ solved, const double* __restrict__ sep, const double* __restrict__ ts, const double* __restrict__ dt,
                        const double* __restrict__ prev_ts, const double* __restrict__ prev_x, double* __restrict__ nodes) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const TrajView& T = C.traj; const pg_vehicle& P = C.veh;
    const double* q0 = state + (size_t)b * 6; const double* u0 = control + (size_t)b * 3;
    const double* TS = ts + (size_t)b * C.NN; const double* DT = dt + (size_t)b * C.N;
    double* ND = nodes + (size_t)b * C.NN * 10;
    double s0 = sep[(size_t)b * 4], e0 = sep[(size_t)b * 4 + 1];
    double E0 = q0[0]; (void)E0;
    double psi0 = q0[2], Ux0 = q0[3], Uy0 = q0[4], r0 = q0[5];
    TrajS tj = traj_at_s(T, s0);                                                   // :76
    double ds = s0 - traj_s_at_time(T, TS[0]);                                     // :77
    double dpsi = adiff(psi0, tj.psi);                                             // :78
    double q[6] = {ds, Ux0, Uy0, r0, dpsi, e0};
    double u[2] = {u0[0], u0[1] + u0[2]};
    double p[2] = {tj.V, tj.kappa};
    auto put = [&](int i) { double* o = ND + i * 10; for (int k = 0; k < 6; k++) o[k] = q[k]; o[6] = u[0]; o[7] = u[1]; o[8] = p[0]; o[9] = p[1]; };
    if (solved[b]) {                                                               // :82-102 with update_interpolations! (:189-195)
        const double* PT = prev_ts + (size_t)b * C.NN; const double* PX = prev_x + (size_t)b * C.NN * 8;
        put(0);
        double tlast = PT[C.NN - 1];
        for (int i = 1; i < C.NN; i++) {
            double t = TS[i];
            double tq = (t < tlast) ? t : tlast;
            int j = clampi(count_leq(PT, C.NN, tq), 1, C.NN - 1) - 1;
            double w = (tq - PT[j]) / (PT[j + 1] - PT[j]);
            for (int k = 0; k < 6; k++) q[k] = (1.0 - w) * PX[j * 8 + k] + w * PX[(j + 1) * 8 + k];
            u[0] = ((1.0 - w) * PX[j * 8 + 6] + w * PX[(j + 1) * 8 + 6]) * C.un0;
            u[1] = ((1.0 - w) * PX[j * 8 + 7] + w * PX[(j + 1) * 8 + 7]) * C.un1;
            double s = traj_s_at_time(T, t) + q[0];                                // :96
            tj = traj_at_s(T, s);
            p[0] = tj.V; p[1] = tj.kappa;
            put(i);
        }
        return;
    }
    // cold start :103-141
    double s = s0, sdp, cdp; sincos(dpsi, &sdp, &cdp);
    double V = Ux0 * cdp - Uy0 * sdp;
    double beta0 = atan2(Uy0, Ux0), delta0 = u0[0];
    double Fyf0, Fyr0;
    {   // lateral_tire_forces(bicycle, q0, u0): raw (delta, Fxf, Fxr), no actuator limits (:110; vehicle_dynamics.jl:78-87)
        double sd, cd; sincos(u0[0], &sd, &cd);
        double af = atan2(Uy0 + P.a * r0, Ux0) - u0[0], ar = atan2(Uy0 - P.b * r0, Ux0);
        lateral_forces<double>(P, af, ar, u0[1], u0[2], sd, cd, Fyf0, Fyr0);
    }
    bool traj_mode = !(toff[b] != toff[b]);
    for (int i = 0; i < C.NN; i++) {
        double tau = (i == C.NN - 1) ? DT[i - 1] : DT[i];
        tj = traj_at_s(T, s);
        ds = s - traj_s_at_time(T, TS[i]);
        double A_des = tj.A + C.cp.k_V * (tj.V - V) / tau + (traj_mode ? -C.cp.k_s * ds / tau / tau : 0.0);
        A_des = jmin(jmax(A_des, (C.cp.V_min - V) / tau), (C.cp.V_max - V) / tau);
        double A;
        if (i == 0) {
            double dUx, dUy, dr;
            world_body_rhs<double>(P, Ux0, Uy0, r0, u0[0], u0[1] + u0[2], dUx, dUy, dr);        // :118
            A = (dUx - r0 * Uy0) * cdp - (dUy + r0 * Ux0) * sdp;                                 // :119
        } else if (i <= C.Ns) {
            Steady est = steady_state(P, V, A_des, tj.kappa, 1, r0, beta0, delta0, Fyf0);        // :122
            q[0] = ds; q[1] = Ux0; q[2] = Uy0; q[3] = r0; q[4] = adiff(psi0, tj.psi); q[5] = e0;
            u[0] = est.delta; u[1] = est.Fx; p[0] = tj.V; p[1] = tj.kappa; A = est.A;
        } else {
            Steady est = steady_state(P, V, A_des, tj.kappa, 4, V * tj.kappa, 0.0, 0.0, 0.0);    // :128
            q[0] = ds; q[1] = est.Ux; q[2] = est.Uy; q[3] = est.r; q[4] = -est.beta; q[5] = 0.0;
            u[0] = est.delta; u[1] = est.Fx; p[0] = tj.V; p[1] = tj.kappa; A = est.A;
        }
        put(i);
        if (i == C.NN - 1) break;
        V = V + A * tau;
        s = s + V * tau + A * tau * tau * 0.5;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// linearize: RK4 (nsub sub-steps) of the tracking model with two tangent directions per lane.
// lane -> (instance, interval t, group g); group g carries tangents {2g, 2g+1} of (q[0..5], u0[0..1], uf[0..1]).
// Writes raw Jacobian columns; group 0 also writes Phi (the propagated state) into the c slot.  k_limits finishes c and scales B.
__global__ __launch_bounds__(256) void k_linearize(DevCfg C, int B, const double* __restrict__ nodes, const double* __restrict__ dt, double* __restrict__ qp) {
    long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long per = (long)C.N * 5;
    if (gid >= (long)B * per) return;
    int b = (int)(gid / per); int rem = (int)(gid - (long)b * per); int t = rem / 5, g = rem - t * 5;
    bool ramp = t >= C.Ns;
    if (!ramp && g == 4) return;
    const double* n0 = nodes + ((size_t)b * C.NN + t) * 10; const double* n1 = n0 + 10;
    double h_total = dt[(size_t)b * C.N + t];
    D2 x[6];
#pragma unroll
    for (int k = 0; k < 6; k++) x[k] = D2(n0[k], (2 * g == k) ? 1.0 : 0.0, (2 * g + 1 == k) ? 1.0 : 0.0);
    D2 u0a(n0[6], g == 3 ? 1.0 : 0.0, 0.0), u0b(n0[7], 0.0, g == 3 ? 1.0 : 0.0);
    D2 ufa(ramp ? n1[6] : n0[6], g == 4 ? 1.0 : 0.0, 0.0), ufb(ramp ? n1[7] : n0[7], 0.0, g == 4 ? 1.0 : 0.0);
    double pV0 = n0[8], pK0 = n0[9], pV1 = ramp ? n1[8] : n0[8], pK1 = ramp ? n1[9] : n0[9];
    const int nsub = C.nsub; const double h = h_total / nsub;
    auto rhs = [&](const D2* xx, double tau, D2* out) {
        double w = ramp ? tau / h_total : 0.0;
        D2 ua = u0a + (ufa - u0a) * w, ub = u0b + (ufb - u0b) * w;
        tracking_rhs<D2>(C.veh, xx, ua, ub, pV0 + (pV1 - pV0) * w, pK0 + (pK1 - pK0) * w, out);
    };
#pragma unroll 1
    for (int i = 0; i < nsub; i++) {
        double t0 = i * h;
        D2 k1[6], k2[6], xx[6], acc[6];
        rhs(x, t0, k1);
#pragma unroll
        for (int k = 0; k < 6; k++) { xx[k] = x[k] + k1[k] * (h * 0.5); acc[k] = k1[k]; }
        rhs(xx, t0 + h * 0.5, k2);
#pragma unroll
        for (int k = 0; k < 6; k++) { xx[k] = x[k] + k2[k] * (h * 0.5); acc[k] = acc[k] + 2.0 * k2[k]; }
        rhs(xx, t0 + h * 0.5, k1);
#pragma unroll
        for (int k = 0; k < 6; k++) { xx[k] = x[k] + k1[k] * h; acc[k] = acc[k] + 2.0 * k1[k]; }
        rhs(xx, t0 + h, k2);
#pragma unroll
        for (int k = 0; k < 6; k++) x[k] = x[k] + (acc[k] + k2[k]) * (h / 6.0);
    }
    QpOff o = qp_offsets(C.N);
    double* Q = qp + (size_t)b * C.qp_len;
    if (g < 3) {
#pragma unroll
        for (int i = 0; i < 6; i++) { Q[o.A + 36 * t + 6 * i + 2 * g] = x[i].a; Q[o.A + 36 * t + 6 * i + 2 * g + 1] = x[i].b; }
    } else if (g == 3) {
#pragma unroll
        for (int i = 0; i < 6; i++) { Q[o.B0 + 12 * t + 2 * i] = x[i].a; Q[o.B0 + 12 * t + 2 * i + 1] = x[i].b; }
    } else {
#pragma unroll
        for (int i = 0; i < 6; i++) { Q[o.Bf + 12 * t + 2 * i] = x[i].a; Q[o.Bf + 12 * t + 2 * i + 1] = x[i].b; }
    }
    if (g == 0) {
#pragma unroll
        for (int i = 0; i < 6; i++) Q[o.c + 6 * t + i] = x[i].v;
    }
}

// finishes c = Phi - A q - B0 u0 - Bf uf, scales B by u_normalization, stability envelope + bounds (:354-367), q_curr/u_curr (:332-333), HJI row (:345-346)
__global__ void k_limits(DevCfg C, int B, const double* __restrict__ nodes, const double* __restrict__ dt, const double* __restrict__ hji_Mb, double* __restrict__ qp) {
    long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long)B * C.N) return;
    int b = (int)(gid / C.N), t = (int)(gid - (long)b * C.N);
    bool ramp = t >= C.Ns;
    const double* n0 = nodes + ((size_t)b * C.NN + t) * 10; const double* n1 = n0 + 10;
    QpOff o = qp_offsets(C.N);
    double* Q = qp + (size_t)b * C.qp_len;
    double* A = Q + o.A + 36 * t; double* B0 = Q + o.B0 + 12 * t; double* Bf = Q + o.Bf + 12 * t; double* c = Q + o.c + 6 * t;
    for (int i = 0; i < 6; i++) {
        double ci = c[i];
        for (int j = 0; j < 6; j++) ci -= A[6 * i + j] * n0[j];
        ci -= B0[2 * i] * n0[6] + B0[2 * i + 1] * n0[7];
        if (ramp) ci -= Bf[2 * i] * n1[6] + Bf[2 * i + 1] * n1[7];
        else { Bf[2 * i] = 0.0; Bf[2 * i + 1] = 0.0; }
        c[i] = ci;
        B0[2 * i] *= C.un0; B0[2 * i + 1] *= C.un1; Bf[2 * i] *= C.un0; Bf[2 * i + 1] *= C.un1;      // :338,350-351
    }
    double Uxt = n1[1], Fx = n1[7];                                                                   // :357-358
    double Fxf = Fx > 0.0 ? Fx * C.veh.fwd_frac : Fx * C.veh.fwb_frac, Fxr = Fx > 0.0 ? Fx * C.veh.rwd_frac : Fx * C.veh.rwb_frac;
    Envelope e = stable_limits(C.veh, Uxt, Fxf, Fxr);
    for (int i = 0; i < 4; i++) { Q[o.H + 8 * t + 2 * i] = e.H[i][0]; Q[o.H + 8 * t + 2 * i + 1] = e.H[i][1]; Q[o.G + 4 * t + i] = e.G[i]; }
    double h = dt[(size_t)b * C.N + t];
    Q[o.dmin + t] = jmax(e.dmin, -C.veh.delta_max) / C.un0;
    Q[o.dmax + t] = jmin(e.dmax, C.veh.delta_max) / C.un0;
    Q[o.fxmax + t] = jmin(C.veh.Px_max / Uxt, C.veh.Fx_max) / C.un1;
    Q[o.ddmin + t] = -C.cp.deltadot_max * h / C.un0;
    Q[o.ddmax + t] = C.cp.deltadot_max * h / C.un0;
    Q[o.dt + t] = h;
    if (t == 0) {
        for (int k = 0; k < 6; k++) Q[o.qcurr + k] = n0[k];
        Q[o.ucurr] = n0[6] / C.un0; Q[o.ucurr + 1] = n0[7] / C.un1;
        if (C.has_hji) { Q[o.M] = hji_Mb[(size_t)b * 4]; Q[o.M + 1] = hji_Mb[(size_t)b * 4 + 1]; Q[o.b] = hji_Mb[(size_t)b * 4 + 2]; }
        else { Q[o.M] = 0.0; Q[o.M + 1] = 0.0; Q[o.b] = 1.0; }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// HJI grid on device: node record = 8 floats (V, gradV[0..6]) -> one 32 B aligned read per corner, 64 B per dim-1 pair.
struct HjiView { int dims[7]; int koff[7]; long stride[7]; const float* knots; const float* nodes; };

// HJIRelativeState(us, them): HJI_computation.jl:20-24 (cpsi = sin(-psi), spsi = cos(-psi): names swapped in the reference)
__global__ void k_hji_relstate(int B, const double* __restrict__ state, const double* __restrict__ other, double* __restrict__ x7) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const double* us = state + (size_t)b * 6; const double* th = other + (size_t)b * 4; double* x = x7 + (size_t)b * 7;
    double s, c; sincos(-us[2], &s, &c);
    double cpsi = s, spsi = c, dE = th[0] - us[0], dN = th[1] - us[1];
    x[0] = cpsi * dE + spsi * dN; x[1] = -spsi * dE + cpsi * dN; x[2] = adiff(th[2], us[2]);
    x[3] = us[3]; x[4] = us[4]; x[5] = th[3]; x[6] = us[5];
}

// cache[x]: HJI_computation.jl:66-72.  One wave per lookup; lane c (6 bits = corner bits of dims 2..7) gathers the dim-1 PAIR
// (2 node records = 64 contiguous bytes), weights in fp64, wave reduction of the 8 channels.  out8[b] = (V, gradV[0..6]).
__global__ __launch_bounds__(256) void k_hji_lookup(HjiView Hv, int B, const double* __restrict__ x7, double* __restrict__ out8) {
    int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= B) return;
    const double* x = x7 + (size_t)wave * 7;
    // lanes 0..6: knot search for their dimension (searchsortedlast, clamp to [1, n-1]) and the in-bounds test (:67)
    int myidx = 0; double myw = 0.0; int inb = 1;
    if (lane < 7) {
        const float* k = Hv.knots + Hv.koff[lane]; int n = Hv.dims[lane]; double xv = x[lane];
        inb = ((double)k[0] <= xv) && (xv <= (double)k[n - 1]);
        int lo = 0, hi = n;
        while (lo < hi) { int mid = (lo + hi) >> 1; if ((double)k[mid] <= xv) lo = mid + 1; else hi = mid; }
        int i = lo < 1 ? 1 : (lo > n - 1 ? n - 1 : lo);
        myidx = i - 1;
        double k0 = k[i - 1], k1 = k[i];
        myw = (xv - k0) / (k1 - k0);
    }
    int all_in = __all(inb);
    double acc[8];
    if (all_in) {
        long off = 0; double wt = 1.0; double w0 = __shfl(myw, 0); int i0 = __shfl(myidx, 0);
        off = (long)i0 * Hv.stride[0];
#pragma unroll
        for (int d = 1; d < 7; d++) {
            double wd = __shfl(myw, d); int id = __shfl(myidx, d);
            int bit = (lane >> (d - 1)) & 1;
            wt *= bit ? wd : (1.0 - wd);
            off += (long)(id + bit) * Hv.stride[d];
        }
        const float4* p = reinterpret_cast<const float4*>(Hv.nodes + off * 8);
        float4 a0 = p[0], a1 = p[1], b0 = p[2], b1 = p[3];
        double wa = wt * (1.0 - w0), wb = wt * w0;
        acc[0] = wa * (double)a0.x + wb * (double)b0.x; acc[1] = wa * (double)a0.y + wb * (double)b0.y;
        acc[2] = wa * (double)a0.z + wb * (double)b0.z; acc[3] = wa * (double)a0.w + wb * (double)b0.w;
        acc[4] = wa * (double)a1.x + wb * (double)b1.x; acc[5] = wa * (double)a1.y + wb * (double)b1.y;
        acc[6] = wa * (double)a1.z + wb * (double)b1.z; acc[7] = wa * (double)a1.w + wb * (double)b1.w;
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) {
#pragma unroll
            for (int k = 0; k < 8; k++) acc[k] += __shfl_xor(acc[k], s);
        }
    } else {
        acc[0] = INFINITY;
#pragma unroll
        for (int k = 1; k < 8; k++) acc[k] = 0.0;
    }
    if (lane < 8) out8[(size_t)wave * 8 + lane] = acc[lane];
}

// optimal_disturbance (dMode=:min) HJI_computation.jl:90-131 + compute_reachability_constraint :160-170, lane = instance
__global__ void k_hji_constraint(DevCfg C, int B, const double* __restrict__ x7, const double* __restrict__ vg8, const double* __restrict__ control,
                                 double* __restrict__ Mb) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const pg_vehicle& P = C.veh;
    const double* x = x7 + (size_t)b * 7; const double* vg = vg8 + (size_t)b * 8; const double* g = vg + 1;
    double* o = Mb + (size_t)b * 4;
    double Vv = vg[0];
    o[3] = Vv;
    if (Vv > C.hji_eps) { o[0] = 0.0; o[1] = 0.0; o[2] = 1.0; return; }           // :163-164
    double uH0, uH1;
    {
        double Ax_max = P.Fx_max / P.m, Pmx = P.Px_max / P.m, maxA = 0.9 * P.mu * P.G;
        double Vh = x[5], lam_Ax = g[5], lam_Ay = g[2] / Vh;
        double nrm = (lam_Ax != lam_Ax || lam_Ay != lam_Ay) ? NAN : hypot(lam_Ax, lam_Ay);
        if (nrm < 1e-3) { uH0 = 0.0; uH1 = 0.0; }
        else {
            double desAx = -lam_Ax * maxA / nrm, desAy = -lam_Ay * maxA / nrm;
            double maxAx = jmin(Ax_max, Pmx / Vh), maxAy = P.kappa_max * Vh * Vh;
            if (desAx > maxAx) {
                if (fabs(desAy) < maxAy) maxAy = jmin(maxAy, sqrt(maxA * maxA - maxAx * maxAx));
                uH0 = copysign(maxAy, desAy) / Vh; uH1 = maxAx;
            } else if (fabs(desAy) > maxAy) {
                if (desAx > 0.0) { maxAx = jmin(sqrt(maxA * maxA - maxAy * maxAy), maxAx); uH0 = copysign(maxAy, desAy) / Vh; uH1 = maxAx; }
                else { uH0 = copysign(maxAy, desAy) / Vh; uH1 = -sqrt(maxA * maxA - maxAy * maxAy); }
            } else { uH0 = desAy / Vh; uH1 = maxAx; }
        }
    }
    const double* u = control + (size_t)b * 3;
    double uR0 = u[0], uR1 = u[1] + u[2];
    D2 dUx, dUy, dr;
    world_body_rhs<D2>(P, x[3], x[4], x[6], D2(uR0, 1.0, 0.0), D2(uR1, 0.0, 1.0), dUx, dUy, dr);   // relative_dynamics :77
    double s, c; sincos(x[2], &s, &c);
    double f0 = x[5] * c - x[3] + x[1] * x[6], f1 = x[5] * s - x[4] - x[0] * x[6], f2 = uH0 - x[6];
    D2 Hm = g[3] * dUx + g[4] * dUy + g[6] * dr + (g[0] * f0 + g[1] * f1 + g[2] * f2 + g[5] * uH1);
    double M0 = Hm.a, M1 = Hm.b;
    o[0] = M0 * C.un0; o[1] = M1 * C.un1;                                          // coupled_lat_long.jl:345
    o[2] = Hm.v - (M0 * uR0 + M1 * uR1);                                           // :168
}

// ------------------------------------------------------------------------------------------------------------------
// QP solve: one wavefront per instance.  See tools/ipm_prototype.py for the algorithm statement and DESIGN.md for the derivation.
// State x_k = (q_k, u_k) in R^8, input v_k = u_{k+1} - u_k; 16 inequality rows per transition k (node k+1):
//   0: Ux >= V_min   1: Ux <= V_max   2: Fx >= Fx_min   3: delta <= dmax   4: delta >= dmin   5: Fx <= fxmax
//   6..9: H_i [Uy;r] - sigma_{i/2} <= G_i     10: sigma1 >= 0   11: sigma2 >= 0    12: d_delta <= ddmax   13: d_delta >= ddmin
//   14: M u + b + sigma_HJI >= 0              15: sigma_HJI >= 0        (14,15 only for nodes 1 .. min(N_HJI,Ns)-1)
struct SolveOut { double* sol_x; double* sol_sigma; double* u_out; int* status; int* iters; uint16_t* active; double* mu; int* solved; };

#define NROW 16

struct StageRows {
    double t[NROW], lam[NROW], corr[NROW];
};

PG_DEV double wave_min(double v) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) { double o = __shfl_xor(v, s); v = o < v ? o : v; }
    return v;
}
PG_DEV double wave_max(double v) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) { double o = __shfl_xor(v, s); v = o > v ? o : v; }
    return v;
}
PG_DEV double wave_sum(double v) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) v += __shfl_xor(v, s);
    return v;
}

__global__ __launch_bounds__(64) void k_solve(DevCfg C, int B, const double* __restrict__ qp, SolveOut O) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int N = C.N, NN = C.NN;
    extern __shared__ double lds[];
    double* sA = lds;                    // [N][36]
    double* sB0 = sA + 36 * N;           // [N][12]
    double* sBf = sB0 + 12 * N;          // [N][12]
    double* sc = sBf + 12 * N;           // [N][6]
    double* sQ = sc + 6 * N;             // [NN][14]: diag[8], (yy, yr, rr), (dd, df, ff)
    double* sq = sQ + 14 * NN;           // [NN][8]
    double* sR = sq + 8 * NN;            // [N][2]   diagonal of Rhat
    double* sr = sR + 2 * N;             // [N][2]
    double* sK = sr + 2 * N;             // [N][2][8]
    double* sSi = sK + 16 * N;           // [N][4]   (Sinv00, Sinv01, Sinv11, -)
    double* sMc = sSi + 4 * N;           // [N][8]   P_{k+1} cbar_k
    double* skf = sMc + 8 * N;           // [N][2]
    double* sx = skf + 2 * N;            // [NN][8]  Newton point
    double* sv = sx + 8 * NN;            // [N][2]
    double* sP = sv + 2 * N;             // [8][8]
    double* sMA = sP + 64;               // [8][8]
    double* sMB = sMA + 64;              // [8][2]
    double* sF = sMB + 16;               // [2][8]
    double* sp = sF + 16;                // [8]  p_{k+1}
    double* sx0 = sp + 8;                // [8]

    const QpOff o = qp_offsets(N);
    const double* Q = qp + (size_t)b * C.qp_len;
    // ---- stage the dynamics blocks in LDS (coalesced: lanes read consecutive doubles) ----
    for (int i = lane; i < 66 * N; i += 64) lds[i] = Q[i];          // A, B0, Bf, c are the first 66 N doubles of the block
    if (lane < 8) sx0[lane] = lane < 6 ? Q[o.qcurr + lane] : Q[o.ucurr + lane - 6];

    // ---- per-stage constants in the registers of lane s (stage s = transition s, node s+1) ----
    const bool act = lane < N;
    const int s = act ? lane : 0;
    double h0[4], h1[4], bb[NROW];
    const double M0 = Q[o.M], M1 = Q[o.M + 1];
    const bool hji_on = act && (s + 1 < (C.cp.N_HJI < C.Ns ? C.cp.N_HJI : C.Ns));
    const double dts = Q[o.dt + s];
    const double Qd0 = 2.0 * C.cp.Q_ds * dts, Qd4 = 2.0 * C.cp.Q_dpsi * dts, Qd5 = 2.0 * C.cp.Q_e * dts, Qd6 = 2.0 * C.cp.R_delta * dts, Qd7 = 2.0 * C.cp.R_Fx * dts;
    const double Rd0 = 2.0 * C.cp.R_ddelta / dts, Rd1 = 2.0 * C.cp.R_dFx / dts;
    const double wb = C.cp.W_beta * dts, wr = C.cp.W_r * dts, wh = C.cp.W_HJI;
#pragma unroll
    for (int i = 0; i < 4; i++) { h0[i] = Q[o.H + 8 * s + 2 * i]; h1[i] = Q[o.H + 8 * s + 2 * i + 1]; bb[6 + i] = Q[o.G + 4 * s + i]; }
    bb[0] = -C.cp.V_min; bb[1] = C.cp.V_max; bb[2] = -C.fxmin_n; bb[3] = Q[o.dmax + s]; bb[4] = -Q[o.dmin + s]; bb[5] = Q[o.fxmax + s];
    bb[10] = 0.0; bb[11] = 0.0; bb[12] = Q[o.ddmax + s]; bb[13] = -Q[o.ddmin + s]; bb[14] = Q[o.b]; bb[15] = 0.0;
    __syncthreads();

    // slack of every row at the point w = (x[8], v0, s1, s2, sh)
    auto slacks = [&](const double* x, double v0, double s1, double s2, double sh, double* out) {
        out[0] = x[1] + bb[0]; out[1] = bb[1] - x[1]; out[2] = x[7] + bb[2]; out[3] = bb[3] - x[6]; out[4] = x[6] + bb[4]; out[5] = bb[5] - x[7];
#pragma unroll
        for (int i = 0; i < 4; i++) out[6 + i] = bb[6 + i] - (h0[i] * x[2] + h1[i] * x[3]) + (i < 2 ? s1 : s2);
        out[10] = s1; out[11] = s2; out[12] = bb[12] - v0; out[13] = v0 + bb[13];
        out[14] = bb[14] + M0 * x[6] + M1 * x[7] + sh; out[15] = sh;
    };
    const int nrows = hji_on ? 16 : 14;

    // forward roll-out through LDS by lanes 0..7: x_{k+1} = Abar x_k + Bbar v_k + cbar  (v from sK/skf when use_gain)
    auto abar = [&](int k, int m, int j) -> double {   // Abar_k[m][j]
        if (m < 6) return j < 6 ? sA[36 * k + 6 * m + j] : sB0[12 * k + 2 * m + (j - 6)] + sBf[12 * k + 2 * m + (j - 6)];
        return m == j ? 1.0 : 0.0;
    };
    auto forward = [&](bool use_gain) {
        if (lane < 8) sx[lane] = sx0[lane];
        __syncthreads();
        for (int k = 0; k < N; k++) {
            double v0 = 0.0, v1 = 0.0;
            if (use_gain) {
                v0 = skf[2 * k]; v1 = skf[2 * k + 1];
#pragma unroll
                for (int m = 0; m < 8; m++) { double xm = sx[8 * k + m]; v0 += sK[16 * k + m] * xm; v1 += sK[16 * k + 8 + m] * xm; }
            }
            if (lane < 8) {
                double acc;
                if (lane < 6) {
                    acc = sc[6 * k + lane] + sBf[12 * k + 2 * lane] * v0 + sBf[12 * k + 2 * lane + 1] * v1;
#pragma unroll
                    for (int m = 0; m < 8; m++) acc += abar(k, lane, m) * sx[8 * k + m];
                } else acc = sx[8 * k + lane] + (lane == 6 ? v0 : v1);
                sx[8 * (k + 1) + lane] = acc;
                if (lane == 0) { sv[2 * k] = v0; sv[2 * k + 1] = v1; }
            }
            __syncthreads();
        }
    };

    // ---- initial point: v = 0 roll-out; sigma just feasible; t = max(slack, tau); lambda = mu0 / t ----
    StageRows R;
    double xs[8], sg1 = 0.0, sg2 = 0.0, sgh = 0.0, vcur0 = 0.0;    // current iterate of this stage: x_{s+1}, sigma, v (v1 is not constrained)
    forward(false);
    double rp0 = 0.0;
    {
#pragma unroll
        for (int m = 0; m < 8; m++) xs[m] = sx[8 * (s + 1) + m];
        double sl[NROW];
        slacks(xs, 0.0, 0.0, 0.0, 0.0, sl);
        const double sig0 = 0.1, tau = 1e-4;
        sg1 = fmax(0.0, -fmin(sl[6], sl[7])) + sig0; sg2 = fmax(0.0, -fmin(sl[8], sl[9])) + sig0; sgh = hji_on ? fmax(0.0, -sl[14]) + sig0 : 0.0;
        slacks(xs, 0.0, sg1, sg2, sgh, sl);
#pragma unroll
        for (int j = 0; j < NROW; j++) {
            bool on = act && j < nrows;
            double tj = on ? fmax(sl[j], tau) : 1.0;
            R.t[j] = tj; R.lam[j] = on ? C.ipm_mu0 / tj : 0.0; R.corr[j] = 0.0;
            if (on) rp0 = fmax(rp0, tj - sl[j]);
        }
    }
    rp0 = wave_max(rp0);
    const double ntot = wave_sum(act ? (double)nrows : 0.0);
    double vcur1 = 0.0;
    double phi = 1.0, mu = 0.0;
    int it = 0, status = PG_MAX_ITER;

    // assemble stage s of the Newton LQ problem from W = lam/t and ell, eliminate the slacks, publish to LDS.
    // returns the elimination data needed to recover sigma+ (d, c, g for sigma1, sigma2, sigma_HJI)
    double e_d1, e_c10, e_c11, e_g1, e_d2, e_c20, e_c21, e_g2, e_dh, e_ch0, e_ch1, e_gh;
    auto assemble = [&](double sigmu, bool matrices) {
        double W[NROW], ell[NROW];
#pragma unroll
        for (int j = 0; j < NROW; j++) {
            bool on = j < nrows;
            W[j] = on ? R.lam[j] / R.t[j] : 0.0;
            ell[j] = on ? (sigmu - R.corr[j]) / R.t[j] + R.lam[j] - W[j] * bb[j] : 0.0;
        }
        double g1 = -ell[0] + ell[1], g7 = -ell[2] + ell[5] - M1 * ell[14], g6 = ell[3] - ell[4] - M0 * ell[14];
        double g2 = 0.0, g3 = 0.0;
#pragma unroll
        for (int i = 0; i < 4; i++) { g2 += h0[i] * ell[6 + i]; g3 += h1[i] * ell[6 + i]; }
        e_g1 = wb - ell[6] - ell[7] - ell[10]; e_g2 = wr - ell[8] - ell[9] - ell[11]; e_gh = wh - ell[14] - ell[15];
        double gv0 = ell[12] - ell[13];
        e_d1 = W[6] + W[7] + W[10]; e_d2 = W[8] + W[9] + W[11]; e_dh = hji_on ? W[14] + W[15] : 1.0;
        e_c10 = -(W[6] * h0[0] + W[7] * h0[1]); e_c11 = -(W[6] * h1[0] + W[7] * h1[1]);
        e_c20 = -(W[8] * h0[2] + W[9] * h0[3]); e_c21 = -(W[8] * h1[2] + W[9] * h1[3]);
        e_ch0 = W[14] * M0; e_ch1 = W[14] * M1;
        if (!hji_on) e_gh = 0.0;
        if (act) {
            double* qo = sq + 8 * (s + 1);
            qo[0] = 0.0; qo[1] = g1; qo[4] = 0.0; qo[5] = 0.0;
            qo[2] = g2 - e_c10 * e_g1 / e_d1 - e_c20 * e_g2 / e_d2;
            qo[3] = g3 - e_c11 * e_g1 / e_d1 - e_c21 * e_g2 / e_d2;
            qo[6] = g6 - e_ch0 * e_gh / e_dh; qo[7] = g7 - e_ch1 * e_gh / e_dh;
            sr[2 * s] = gv0; sr[2 * s + 1] = 0.0;
            if (matrices) {
                double* Qo = sQ + 14 * (s + 1);
                Qo[0] = Qd0; Qo[1] = W[0] + W[1]; Qo[2] = 0.0; Qo[3] = 0.0; Qo[4] = Qd4; Qo[5] = Qd5;
                Qo[6] = Qd6 + W[3] + W[4] + M0 * M0 * W[14] - e_ch0 * e_ch0 / e_dh;
                Qo[7] = Qd7 + W[2] + W[5] + M1 * M1 * W[14] - e_ch1 * e_ch1 / e_dh;
                Qo[13] = 0.0;   // (ff) slot unused: diag carries it
                double yy = 0.0, yr = 0.0, rr = 0.0;
#pragma unroll
                for (int i = 0; i < 4; i++) { yy += W[6 + i] * h0[i] * h0[i]; yr += W[6 + i] * h0[i] * h1[i]; rr += W[6 + i] * h1[i] * h1[i]; }
                Qo[8] = yy - e_c10 * e_c10 / e_d1 - e_c20 * e_c20 / e_d2;
                Qo[9] = yr - e_c10 * e_c11 / e_d1 - e_c20 * e_c21 / e_d2;
                Qo[10] = rr - e_c11 * e_c11 / e_d1 - e_c21 * e_c21 / e_d2;
                Qo[11] = M0 * M1 * W[14] - e_ch0 * e_ch1 / e_dh;      // (delta, Fx) off-diagonal
                Qo[12] = 0.0;
                sR[2 * s] = Rd0 + W[12] + W[13]; sR[2 * s + 1] = Rd1;
            }
        }
    };
    // Qhat_k[i][j] from the packed per-node record (node k >= 1; node 0 has no cost)
    auto qhat = [&](int k, int i, int j) -> double {
        const double* Qo = sQ + 14 * k;
        if (i == j) return (i == 2) ? Qo[8] : (i == 3) ? Qo[10] : Qo[i];
        if ((i == 2 && j == 3) || (i == 3 && j == 2)) return Qo[9];
        if ((i == 6 && j == 7) || (i == 7 && j == 6)) return Qo[11];
        return 0.0;
    };

    const int li = lane >> 3, lj = lane & 7;

    // Riccati matrix pass (once per IPM iteration): lane (li, lj) owns P[li][lj]
    auto riccati_matrices = [&]() {
        double Pij = qhat(N, li, lj);
        for (int k = N - 1; k >= 0; k--) {
            sP[lane] = Pij;
            __syncthreads();
            double ma = 0.0, second = 0.0;
#pragma unroll
            for (int m = 0; m < 8; m++) {
                double pim = sP[8 * li + m];
                ma += pim * abar(k, m, lj);
                double col = 0.0;
                if (lj < 2) col = m < 6 ? sBf[12 * k + 2 * m + lj] : ((m - 6) == lj ? 1.0 : 0.0);
                else if (lj == 2) col = m < 6 ? sc[6 * k + m] : 0.0;
                second += pim * col;
            }
            sMA[lane] = ma;
            if (lj < 2) sMB[2 * li + lj] = second;
            if (lj == 2) sMc[8 * k + li] = second;
            __syncthreads();
            // F[c][lj] = Bbar' MA ; S = Rhat + Bbar' MB
            double F0 = sMA[8 * 6 + lj], F1 = sMA[8 * 7 + lj];
            double S00 = sR[2 * k] + sMB[2 * 6 + 0], S01 = sMB[2 * 6 + 1], S11 = sR[2 * k + 1] + sMB[2 * 7 + 1];
            double macol[8];
#pragma unroll
            for (int m = 0; m < 8; m++) macol[m] = sMA[8 * m + lj];
#pragma unroll
            for (int m = 0; m < 6; m++) {
                double b0 = sBf[12 * k + 2 * m], b1 = sBf[12 * k + 2 * m + 1];
                F0 += b0 * macol[m]; F1 += b1 * macol[m];
                S00 += b0 * sMB[2 * m]; S01 += b0 * sMB[2 * m + 1]; S11 += b1 * sMB[2 * m + 1];
            }
            double det = S00 * S11 - S01 * S01, idet = 1.0 / det;
            double I00 = S11 * idet, I01 = -S01 * idet, I11 = S00 * idet;
            double K0 = -(I00 * F0 + I01 * F1), K1 = -(I01 * F0 + I11 * F1);
            if (li == 0) { sK[16 * k + lj] = K0; sK[16 * k + 8 + lj] = K1; sF[lj] = F0; sF[8 + lj] = F1; }
            if (lane == 0) { sSi[4 * k] = I00; sSi[4 * k + 1] = I01; sSi[4 * k + 2] = I11; }
            __syncthreads();
            if (k > 0) {
                double pn = qhat(k, li, lj) + sF[li] * K0 + sF[8 + li] * K1;
#pragma unroll
                for (int m = 0; m < 8; m++) pn += abar(k, m, li) * macol[m];
                // symmetrise through LDS
                sP[lane] = pn;
                __syncthreads();
                Pij = 0.5 * (pn + sP[8 * lj + li]);
                __syncthreads();
            }
        }
    };
    // Riccati vector pass backward (lanes 0..7): p_k = qhat_k + Abar'(y) + K' f,  y = Mc_k + p_{k+1},  f = rhat + Bbar' y,  kff = -Sinv f
    auto riccati_vectors = [&]() {
        if (lane < 8) sp[lane] = sq[8 * N + lane];
        __syncthreads();
        for (int k = N - 1; k >= 0; k--) {
            double y[8];
#pragma unroll
            for (int m = 0; m < 8; m++) y[m] = sMc[8 * k + m] + sp[m];
            double f0 = sr[2 * k] + y[6], f1 = sr[2 * k + 1] + y[7];
#pragma unroll
            for (int m = 0; m < 6; m++) { f0 += sBf[12 * k + 2 * m] * y[m]; f1 += sBf[12 * k + 2 * m + 1] * y[m]; }
            double pn = 0.0;
            if (lane < 8 && k > 0) {
                pn = sq[8 * k + lane] + sK[16 * k + lane] * f0 + sK[16 * k + 8 + lane] * f1;
#pragma unroll
                for (int m = 0; m < 8; m++) pn += abar(k, m, lane) * y[m];
            }
            __syncthreads();
            if (lane < 8) sp[lane] = pn;
            if (lane == 0) {
                double I00 = sSi[4 * k], I01 = sSi[4 * k + 1], I11 = sSi[4 * k + 2];
                skf[2 * k] = -(I00 * f0 + I01 * f1); skf[2 * k + 1] = -(I01 * f0 + I11 * f1);
            }
            __syncthreads();
        }
    };
    // Newton point of this stage from LDS + slack recovery; computes t+ per row
    double xn[8], vn0, vn1, sn1, sn2, snh;
    auto newton_point = [&](double* tplus) {
#pragma unroll
        for (int m = 0; m < 8; m++) xn[m] = sx[8 * (s + 1) + m];
        vn0 = sv[2 * s]; vn1 = sv[2 * s + 1];
        sn1 = -(e_c10 * xn[2] + e_c11 * xn[3] + e_g1) / e_d1;
        sn2 = -(e_c20 * xn[2] + e_c21 * xn[3] + e_g2) / e_d2;
        snh = hji_on ? -(e_ch0 * xn[6] + e_ch1 * xn[7] + e_gh) / e_dh : 0.0;
        slacks(xn, vn0, sn1, sn2, snh, tplus);
    };

    for (it = 0; it < C.ipm_max_iter; it++) {
        double musum = 0.0;
#pragma unroll
        for (int j = 0; j < NROW; j++) musum += (act && j < nrows) ? R.t[j] * R.lam[j] : 0.0;
        mu = wave_sum(musum) / ntot;
        if (!(mu == mu) || fabs(mu) > 1e300) { status = PG_NUMERICAL; break; }
        if (mu <= C.ipm_tol && phi * fmax(rp0, 1.0) <= C.ipm_tol) { status = PG_SOLVED; break; }

        // ---- predictor (sigma = 0, no correction) ----
#pragma unroll
        for (int j = 0; j < NROW; j++) R.corr[j] = 0.0;
        assemble(0.0, true);
        __syncthreads();
        riccati_matrices();
        riccati_vectors();
        forward(true);
        double tp[NROW], dta[NROW], dla[NROW];
        newton_point(tp);
        double amax = 1.0;
#pragma unroll
        for (int j = 0; j < NROW; j++) {
            bool on = act && j < nrows;
            double W = R.lam[j] / R.t[j];
            dta[j] = tp[j] - R.t[j];
            dla[j] = -W * tp[j];                          // lambda+ - lambda with sigma*mu = 0, corr = 0
            if (on && dta[j] < 0.0) amax = fmin(amax, -R.t[j] / dta[j]);
            if (on && dla[j] < 0.0) amax = fmin(amax, -R.lam[j] / dla[j]);
        }
        double aaff = wave_min(amax);
        double msum = 0.0;
#pragma unroll
        for (int j = 0; j < NROW; j++) msum += (act && j < nrows) ? (R.t[j] + aaff * dta[j]) * (R.lam[j] + aaff * dla[j]) : 0.0;
        double mu_aff = wave_sum(msum) / ntot;
        double sg = mu_aff / mu; sg = sg * sg * sg;
        // ---- corrector ----
#pragma unroll
        for (int j = 0; j < NROW; j++) R.corr[j] = dta[j] * dla[j];
        assemble(sg * mu, false);
        __syncthreads();
        riccati_vectors();
        forward(true);
        newton_point(tp);
        amax = 1e300;
#pragma unroll
        for (int j = 0; j < NROW; j++) {
            bool on = act && j < nrows;
            double W = R.lam[j] / R.t[j];
            dta[j] = tp[j] - R.t[j];
            dla[j] = (sg * mu - R.corr[j]) / R.t[j] - W * tp[j];
            if (on && dta[j] < 0.0) amax = fmin(amax, -R.t[j] / dta[j]);
            if (on && dla[j] < 0.0) amax = fmin(amax, -R.lam[j] / dla[j]);
        }
        double alpha = fmin(1.0, 0.995 * wave_min(amax));
#pragma unroll
        for (int j = 0; j < NROW; j++) {
            bool on = act && j < nrows;
            if (on) { R.t[j] += alpha * dta[j]; R.lam[j] += alpha * dla[j]; }
        }
#pragma unroll
        for (int m = 0; m < 8; m++) xs[m] += alpha * (xn[m] - xs[m]);
        vcur0 += alpha * (vn0 - vcur0); vcur1 += alpha * (vn1 - vcur1);
        sg1 += alpha * (sn1 - sg1); sg2 += alpha * (sn2 - sg2); sgh += alpha * (snh - sgh);
        phi *= (1.0 - alpha);
        __syncthreads();
    }
    if (status == PG_SOLVED) {
        double Ux0 = sx0[1], Fx0 = sx0[7];
        if (Ux0 < C.cp.V_min || Ux0 > C.cp.V_max || Fx0 < C.fxmin_n) status = PG_INFEASIBLE_X0;
    }
    // ---- outputs ----
    double* SX = O.sol_x + (size_t)b * NN * 8;
    if (lane < 8) SX[lane] = sx0[lane];
    if (act) {
#pragma unroll
        for (int m = 0; m < 8; m++) SX[8 * (s + 1) + m] = xs[m];
        double* SG = O.sol_sigma + ((size_t)b * N + s) * 3;
        SG[0] = sg1; SG[1] = sg2; SG[2] = sgh;
        unsigned mask = 0;
#pragma unroll
        for (int j = 0; j < NROW; j++) if (j < nrows && R.lam[j] > R.t[j]) mask |= (1u << j);
        O.active[(size_t)b * N + s] = (uint16_t)mask;
    }
    if (lane == 0) {
        // get_next_control: coupled_lat_long.jl:370-374 (node 2 of the reference = stage lane 0's node)
        double d = xs[6] * C.un0, Fx = xs[7] * C.un1;
        double* U = O.u_out + (size_t)b * 3;
        U[0] = d; U[1] = Fx > 0.0 ? Fx * C.veh.fwd_frac : Fx * C.veh.fwb_frac; U[2] = Fx > 0.0 ? Fx * C.veh.rwd_frac : Fx * C.veh.rwb_frac;
        O.status[b] = status; O.iters[b] = it; O.mu[b] = mu; O.solved[b] = 1;      // model_predictive_control.jl:76: solved = true
    }
}

}  // namespace pg
