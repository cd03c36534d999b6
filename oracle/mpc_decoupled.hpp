// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the shipped product path.
// CPU restatement of /root/reference/src/decoupled_lat_long.jl (all) + the lateral model of src/vehicle_dynamics.jl:185-224.
// Parity UNPINNED (no golden vectors in the reference).  Third-party pieces restated from published behaviour:
//   * `linearize(dynamics, x, u)` (LinearDynamicsModels 1.0.0, call sites decoupled_lat_long.jl:172,182,245,253): continuous-time
//     Jacobians A = df/dx, B = df/d[u;p] by forward-mode AD, c = f - A x - B [u;p];
//   * `linearize(::LinearDynamics, x, StepControl/RampControl; keep_control_dims=(1,))` (:173,183,246,254): EXACT zero-/first-order-hold
//     discretisation of x' = A x + B w + c, w(s) = w0 + (s/dt)(wf - w0): x+ = Ad x + (G0 - G1) B w0 + G1 B wf + G0 c with
//     Ad = exp(A dt), G0 = int_0^dt exp(A s) ds, G1 = (1/dt) int_0^dt exp(A (dt - s)) s ds; the kept column (delta) gives B / B0, Bf,
//     everything else folds into c.  Computed here by scaling-and-squaring of the Taylor series (rounding-level exact).
#pragma once
#include <cmath>
#include <vector>
#include "mpc_coupled.hpp"

namespace po {

// decoupled_lat_long.jl:1-30
struct DecoupledControlParams {
    double V_min = 1.0, V_max = 15.0, k_V = 10.0 / 4 / 100, k_s = 10.0 / 4 / 10000, deltadot_max = 0.344;
    double Q_dpsi = 1.0 / ((10 * M_PI / 180) * (10 * M_PI / 180)), Q_e = 1.0, W_beta = 50 / (10 * M_PI / 180), W_r = 50.0;
    double R_delta = 0.0, R_ddelta = 0.01 / ((10 * M_PI / 180) * (10 * M_PI / 180));
};

// VehicleModel{LateralTrackingBicycleModel}: vehicle_dynamics.jl:310-316 over :205-224.  q = (Uy, r, dpsi, e), u = (delta, Fx), p = (Ux, kappa, theta, phi)
template <class T>
inline void vehicle_lateral_dynamics(const VehicleParams& P, const T q[4], const T u[2], const T p[4], T out[4]) {
    T d, Fx, Fxf, Fxr;
    apply_control_limits<T>(P, u[0], u[1], value(p[0]), d, Fx);          // get_Ux = p[1] (:309), by value (:295)
    longitudinal_tire_forces<T>(P, Fx, Fxf, Fxr);
    T Ux = p[0];
    T s = sin(q[2]), c = cos(q[2]), sd = sin(d), cd = cos(d);
    T af = atan2(q[0] + P.a * q[1], Ux) - d;
    T ar = atan2(q[0] - P.b * q[1], Ux);
    T Fyf, Fyr;
    lateral_tire_forces<T>(P, af, ar, Fxf, Fxr, sd, cd, Fyf, Fyr);
    T Fyf_t = Fyf * cd + Fxf * sd;
    out[0] = (Fyf_t + Fyr) / P.m - q[1] * Ux;
    out[1] = (P.a * Fyf_t - P.b * Fyr) / P.Izz;
    out[2] = q[1] - Ux * p[1];
    out[3] = Ux * s + q[0] * c;
}

// small dense helpers (4x4, row-major)
struct M4 { double a[16]; };
inline M4 m4_mul(const M4& x, const M4& y) { M4 r; for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { double s = 0; for (int k = 0; k < 4; k++) s += x.a[4 * i + k] * y.a[4 * k + j]; r.a[4 * i + j] = s; } return r; }
inline M4 m4_add(const M4& x, const M4& y) { M4 r; for (int i = 0; i < 16; i++) r.a[i] = x.a[i] + y.a[i]; return r; }
inline M4 m4_scale(const M4& x, double s) { M4 r; for (int i = 0; i < 16; i++) r.a[i] = x.a[i] * s; return r; }
inline M4 m4_eye() { M4 r; for (int i = 0; i < 16; i++) r.a[i] = 0; for (int i = 0; i < 4; i++) r.a[5 * i] = 1; return r; }
// Ad = exp(A T), G0 = int_0^T exp(A s) ds, G2 = int_0^T G0(s) ds  (G1 of the header = G2 / T)
inline void expm_integrals(const M4& A, double T, M4& Ad, M4& G0, M4& G2) {
    double nrm = 0; for (int i = 0; i < 4; i++) { double s = 0; for (int j = 0; j < 4; j++) s += std::fabs(A.a[4 * i + j]); nrm = std::max(nrm, s); }
    int sq = 0; double h = T;
    while (nrm * h > 0.25 && sq < 40) { h *= 0.5; sq++; }
    M4 Ah = m4_scale(A, h), term = m4_eye();
    Ad = m4_eye(); G0 = m4_scale(m4_eye(), h); G2 = m4_scale(m4_eye(), h * h / 2);
    for (int k = 1; k <= 16; k++) {        // term = (A h)^k / k!
        term = m4_scale(m4_mul(term, Ah), 1.0 / k);
        Ad = m4_add(Ad, term);
        G0 = m4_add(G0, m4_scale(term, h / (k + 1)));
        G2 = m4_add(G2, m4_scale(term, h * h / ((k + 1.0) * (k + 2.0))));
    }
    for (int i = 0; i < sq; i++) {          // doubling: G2(2h) = G2 + h G0 + Ad G2; G0(2h) = G0 + Ad G0; Ad(2h) = Ad Ad
        G2 = m4_add(m4_add(G2, m4_scale(G0, h)), m4_mul(Ad, G2));
        G0 = m4_add(G0, m4_mul(Ad, G0));
        Ad = m4_mul(Ad, Ad);
        h *= 2;
    }
}

struct StageDataDec {
    int Ns = 0, Nl = 0;
    std::vector<double> A, B0, Bf, c, H, G, dmin, dmax, ddmin, ddmax, dt;     // [k][4][4], [k][4], [k][4], [k][4], [k][4][2], [k][4], [k]...
    double q_curr[4], d_curr;
    void resize(int ns, int nl) {
        Ns = ns; Nl = nl; int N = ns + nl;
        A.assign(16 * N, 0); B0.assign(4 * N, 0); Bf.assign(4 * N, 0); c.assign(4 * N, 0); H.assign(8 * N, 0); G.assign(4 * N, 0);
        dmin.assign(N, 0); dmax.assign(N, 0); ddmin.assign(N, 0); ddmax.assign(N, 0); dt.assign(N, 0);
    }
};
struct NodesDec { std::vector<double> qs, us, ps, edges; };    // [i][4], [i][2], [i][4]; edges [i][2] = (edge_L, edge_R) of the tube at the node (trajectories.jl:19-20)

struct DecoupledMPC {
    VehicleParams veh = X1();
    DecoupledControlParams cp;
    MPCTimeSteps TS;
    TrajectoryTube traj;
    void init(int Ns, int Nl, double dts, double dtl, bool corr) { TS.N_short = Ns; TS.N_long = Nl; TS.dt_short = dts; TS.dt_long = dtl; TS.use_correction_step = corr; TS.init(); }
    int N() const { return TS.N_short + TS.N_long; }

    // compute_linearization_nodes!: decoupled_lat_long.jl:52-104 (no warm branch in the reference)
    void linearization_nodes(const double q0[6], const double u0[3], double time_offset, NodesDec& out) const {
        int Ns = TS.N_short, Nn = N() + 1;
        const std::vector<double>&ts = TS.ts, &dt = TS.dt;
        out.qs.assign(4 * Nn, 0); out.us.assign(2 * Nn, 0); out.ps.assign(4 * Nn, 0); out.edges.assign(2 * Nn, 0);
        double s, e0, t0;
        traj.path_coordinates(q0[0], q0[1], s, e0, t0);                                  // :65
        double V = std::hypot(q0[3], q0[4]);                                             // :67
        double beta0 = std::atan2(q0[4], q0[3]), r0 = q0[5], delta0 = u0[0];
        double Fyf0, Fyr0;
        lateral_tire_forces_q(veh, q0[3], q0[4], q0[5], u0[0], u0[1], u0[2], Fyf0, Fyr0); // :71
        for (int i = 0; i < Nn; i++) {
            double tau = (i == Nn - 1) ? dt[i - 1] : dt[i];
            TrajectoryNode tj = traj.at_s(s);
            double kappa = tj.kappa;
            double A_des = tj.A + cp.k_V * (tj.V - V) / tau + (std::isnan(time_offset) ? 0.0 : cp.k_s * (traj.at_time(ts[i]).s - s) / tau / tau);   // :76
            A_des = jl_min(jl_max(A_des, (cp.V_min - V) / tau), (cp.V_max - V) / tau);
            double q[4], u[2], p[4] = {0, kappa, 0, 0}, A;
            if (i == 0) {
                q[0] = q0[4]; q[1] = q0[5]; q[2] = adiff(q0[2], tj.psi); q[3] = e0;      // :79
                u[0] = u0[0]; u[1] = u0[1] + u0[2]; p[0] = q0[3];
                double ud[2] = {u0[0], u0[1] + u0[2]}, qd[6];
                vehicle_world_dynamics<double>(veh, q0, ud, qd);                         // :82
                A = (qd[3] - q0[5] * q0[4]) * std::cos(beta0) + (qd[4] + q0[5] * q0[3]) * std::sin(beta0);   // :83
            } else if (i <= Ns) {
                q[0] = q0[4]; q[1] = q0[5]; q[2] = adiff(q0[2], tj.psi); q[3] = e0;      // :85
                SteadyState est = steady_state_estimates(veh, V, A_des, kappa, 1, r0, beta0, delta0, Fyf0);
                u[0] = est.delta; u[1] = est.Fxf + est.Fxr; p[0] = est.Ux; A = est.A;
            } else {
                SteadyState est = steady_state_estimates(veh, V, A_des, kappa, 4, V * kappa, 0, 0, 0);
                q[0] = est.Uy; q[1] = est.r; q[2] = -est.beta; q[3] = 0;                 // :92
                u[0] = est.delta; u[1] = est.Fxf + est.Fxr; p[0] = est.Ux; A = est.A;
            }
            for (int k = 0; k < 4; k++) { out.qs[4 * i + k] = q[k]; out.ps[4 * i + k] = p[k]; }
            out.us[2 * i] = u[0]; out.us[2 * i + 1] = u[1];
            out.edges[2 * i] = tj.edge_L; out.edges[2 * i + 1] = tj.edge_R;
            if (i == Nn - 1) break;
            V = V + A * tau;
            s = s + V * tau + A * tau * tau / 2;
        }
    }
    // continuous linearisation at (q, w = [u; p]): A (4x4), B (4x6), c (4)
    void continuous(const double* q, const double* w, M4& A, double B[4][6], double c[4]) const {
        typedef Dual<10> D;
        D x[4], u[2], p[4], f[4];
        for (int k = 0; k < 4; k++) x[k] = D::seed(q[k], k);
        for (int k = 0; k < 2; k++) u[k] = D::seed(w[k], 4 + k);
        for (int k = 0; k < 4; k++) p[k] = D::seed(w[2 + k], 6 + k);
        vehicle_lateral_dynamics<D>(veh, x, u, p, f);
        for (int i = 0; i < 4; i++) {
            double ci = f[i].v;
            for (int j = 0; j < 4; j++) { A.a[4 * i + j] = f[i].d[j]; ci -= f[i].d[j] * q[j]; }
            for (int j = 0; j < 6; j++) { B[i][j] = f[i].d[4 + j]; ci -= f[i].d[4 + j] * w[j]; }
            c[i] = ci;
        }
    }
    // one interval: decoupled_lat_long.jl:245-246 (ZOH) / :253-254 (FOH)
    void linearize_interval(const double* q, const double* w0, const double* wf, double dt, bool ramp, double* Ad16, double* B0, double* Bf, double* cd) const {
        M4 A, Ad, G0, G2; double B[4][6], c[4];
        continuous(q, w0, A, B, c);
        expm_integrals(A, dt, Ad, G0, G2);
        M4 G1 = m4_scale(G2, 1.0 / dt);
        for (int i = 0; i < 16; i++) Ad16[i] = Ad.a[i];
        for (int i = 0; i < 4; i++) {
            double b0 = 0, bf = 0, ci = 0;
            for (int k = 0; k < 4; k++) {
                double g0 = G0.a[4 * i + k], g1 = ramp ? G1.a[4 * i + k] : 0.0;
                b0 += (g0 - g1) * B[k][0]; bf += g1 * B[k][0];
                double fold = c[k] * g0;
                for (int j = 1; j < 6; j++) fold += (g0 - g1) * B[k][j] * w0[j] + g1 * B[k][j] * (ramp ? wf[j] : 0.0);
                ci += fold;
            }
            B0[i] = b0; Bf[i] = bf; cd[i] = ci;
        }
    }
    // update_QP!: decoupled_lat_long.jl:228-273
    void update_qp(const NodesDec& nd, StageDataDec& sd) const {
        int Ns = TS.N_short, Nl = TS.N_long, Nt = Ns + Nl;
        sd.resize(Ns, Nl);
        const double *qs = nd.qs.data(), *us = nd.us.data(), *ps = nd.ps.data();
        for (int k = 0; k < Nt; k++) sd.dt[k] = TS.dt[k];
        for (int k = 0; k < 4; k++) sd.q_curr[k] = qs[k];
        sd.d_curr = us[0];
        for (int t = 0; t < Nt; t++) {
            double w0[6] = {us[2 * t], us[2 * t + 1], ps[4 * t], ps[4 * t + 1], ps[4 * t + 2], ps[4 * t + 3]};
            double wf[6] = {us[2 * t + 2], us[2 * t + 3], ps[4 * t + 4], ps[4 * t + 5], ps[4 * t + 6], ps[4 * t + 7]};
            linearize_interval(qs + 4 * t, w0, wf, TS.dt[t], t >= Ns, &sd.A[16 * t], &sd.B0[4 * t], &sd.Bf[4 * t], &sd.c[4 * t]);
        }
        for (int t = 0; t < Nt; t++) {                                                   // :262-272
            double Uxt = ps[4 * (t + 1)], Fxf, Fxr, Fx = us[2 * (t + 1) + 1];
            longitudinal_tire_forces<double>(veh, Fx, Fxf, Fxr);
            StableLimits sl = stable_limits(veh, Uxt, Fxf, Fxr);
            for (int i = 0; i < 4; i++) { sd.H[8 * t + 2 * i] = sl.H[i][0]; sd.H[8 * t + 2 * i + 1] = sl.H[i][1]; sd.G[4 * t + i] = sl.G[i]; }
            sd.dmin[t] = jl_max(sl.delta_min, -veh.delta_max); sd.dmax[t] = jl_min(sl.delta_max, veh.delta_max);
            sd.ddmin[t] = -cp.deltadot_max * TS.dt[t]; sd.ddmax[t] = cp.deltadot_max * TS.dt[t];
        }
    }
    // get_next_control: decoupled_lat_long.jl:275-278 (delta from the QP, Fx from the seeded us[2])
    void next_control(double delta_qp, double Fx_seed, double out[3]) const {
        double Fxf, Fxr; longitudinal_tire_forces<double>(veh, Fx_seed, Fxf, Fxr);
        out[0] = delta_qp; out[1] = Fxf; out[2] = Fxr;
    }
};

}  // namespace po
