// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the shipped product path.
// CPU restatement of /root/reference/src/model_predictive_control.jl:1-30,70-100 and
// src/coupled_lat_long.jl (all).  Parity UNPINNED (reference has no golden vectors; test/runtests.jl:1-5).
//
// Third-party arithmetic that is NOT under /root/reference and is restated here from its published
// behaviour (versions from env/Manifest.toml):
//   * DifferentialDynamicsModels 1.0.0 `propagate` (call site model_predictive_control.jl:94):
//     classical RK4 with a fixed number of sub-steps per control interval (default 10), control sampled
//     at the stage times; RampControl interpolates [u;p] linearly from (u0;p0) at 0 to (uf;pf) at dt.
//   * LinearDynamicsModels 1.0.0 `linearize(f, x, StepControl/RampControl; keep_control_dims)`
//     (call sites coupled_lat_long.jl:253,262,336,348): A = dPhi/dx, B = dPhi/du[keep] (B0, Bf for the
//     ramp end points), c = Phi(x,u) - A x - B u[keep] (- B0 u0 - Bf uf), Jacobians by forward-mode AD
//     THROUGH the integrator (ForwardDiff 0.10.3).
//   The number of RK4 sub-steps is a configuration value of this build (rk4_substeps).
#pragma once
#include "julia_range.hpp"
#include <cmath>
#include <vector>
#include "dual.hpp"
#include "hji.hpp"
#include "trajectory.hpp"
#include "vehicle.hpp"

namespace po {

// coupled_lat_long.jl:1-40
struct CoupledControlParams {
    double V_min = 1.0, V_max = 15.0, k_V = 10.0 / 4 / 100, k_s = 10.0 / 4 / 10000, deltadot_max = 0.344;
    double Q_ds = 1.0, Q_dpsi = 1.0, Q_e = 1.0, W_beta = 50 / (10 * M_PI / 180), W_r = 50.0, W_HJI = 500.0;
    int N_HJI = 3;
    double R_delta = 0.0, R_ddelta = 0.1, R_Fx = 0.0, R_dFx = 0.5;
};

// model_predictive_control.jl:1-30
struct MPCTimeSteps {
    int N_short = 10, N_long = 20;
    double dt_short = 0.01, dt_long = 0.2;
    bool use_correction_step = true;
    // The reference's constructor passes the SAME array as `ts` and `prev_ts` (model_predictive_control.jl:15,
    // `MPCTimeSteps(ts, dt, ..., ts)`), so `TS.prev_ts .= TS.ts` (:20) is a self-copy and prev_ts always equals the
    // CURRENT grid: the warm branch (coupled_lat_long.jl:82-102) therefore re-uses the previous solution node-for-node.
    // alias_prev_ts = true reproduces that behaviour; false gives the (apparently intended) time-shifted interpolation.
    bool alias_prev_ts = true;
    // :25-26 build the grid out of Julia RANGES (`t0 .+ dt_short*(0:N_short)`): julia_range.hpp restates their twice-precision arithmetic (one rounding per element, dt
    // lifted to its rational).  naive_time_grid = true: `t0 + dt*i` with two roundings, the form of rounds 1-5 (kept for A/B and for the knife-edge test)
    bool naive_time_grid = false;
    std::vector<double> ts, dt, prev_ts;
    void init() {
        int N = 1 + N_short + N_long;
        ts.resize(N); dt.resize(N - 1);
        for (int i = 0; i < N; i++) ts[i] = i + 1;          // :13 "initialized so that dt's are nonzero"
        for (int i = 0; i < N - 1; i++) dt[i] = ts[i + 1] - ts[i];
        prev_ts = ts;
    }
    void compute(double t0) {                                 // :17-30
        prev_ts = ts;
        double t0_long = t0 + N_short * dt_short;
        if (use_correction_step) t0_long = dt_long * std::ceil((t0_long + dt_short) / dt_long - 1);
        if (naive_time_grid) {
            for (int i = 0; i <= N_short; i++) ts[i] = t0 + dt_short * i;
            for (int i = 1; i <= N_long; i++) ts[N_short + i] = t0_long + dt_long * i;
        } else {
            const jlrange::Range rs = jlrange::shifted(jlrange::scalar_times_unitrange(dt_short, 0, N_short), t0);            // t0 .+ dt_short*(0:N_short)       :25
            const jlrange::Range rl = jlrange::shifted(jlrange::scalar_times_unitrange(dt_long, 1, N_long), t0_long);         // t0_long .+ dt_long*(1:N_long)    :26
            for (int i = 0; i <= N_short; i++) ts[i] = jlrange::elem(rs, i + 1);
            for (int i = 1; i <= N_long; i++) ts[N_short + i] = jlrange::elem(rl, i);
        }
        for (int i = 0; i < N_short + N_long; i++) dt[i] = ts[i + 1] - ts[i];
        if (alias_prev_ts) prev_ts = ts;
    }
};

// Numeric content of one refreshed QP (what update_QP! writes, coupled_lat_long.jl:315-368), stage-wise.
struct StageData {
    int Ns = 0, Nl = 0;
    std::vector<double> A, B0, Bf, c;          // [k][6][6], [k][6][2] (already scaled by u_normalization), [k][6][2], [k][6]
    std::vector<double> H, G;                  // [k][4][2], [k][4]
    std::vector<double> dmin, dmax, fxmax, ddmin, ddmax, dt;   // [k]
    double q_curr[6], u_curr[2], M_hji[2], b_hji;
    void resize(int ns, int nl) {
        Ns = ns; Nl = nl; int N = ns + nl;
        A.assign(36 * N, 0); B0.assign(12 * N, 0); Bf.assign(12 * N, 0); c.assign(6 * N, 0);
        H.assign(8 * N, 0); G.assign(4 * N, 0);
        dmin.assign(N, 0); dmax.assign(N, 0); fxmax.assign(N, 0); ddmin.assign(N, 0); ddmax.assign(N, 0); dt.assign(N, 0);
    }
};

struct Nodes { std::vector<double> qs, us, ps; };   // [i][6], [i][2], [i][4]

struct CoupledMPC {
    VehicleParams veh = X1();
    CoupledControlParams cp;
    MPCTimeSteps TS;
    TrajectoryTube traj;
    HJICache hji;
    double HJI_eps = 0.05;                   // model_predictive_control.jl:67
    int rk4_substeps = 10;
    double u_norm[2];                        // coupled_lat_long.jl:199

    void init(int Ns, int Nl, double dts, double dtl, bool corr) {
        TS.N_short = Ns; TS.N_long = Nl; TS.dt_short = dts; TS.dt_long = dtl; TS.use_correction_step = corr; TS.init();
        u_norm[0] = veh.delta_max; u_norm[1] = std::max(-veh.Fx_min, veh.Fx_max);
    }
    int N() const { return TS.N_short + TS.N_long; }

    // ---- compute_linearization_nodes!: coupled_lat_long.jl:62-142 ------------------------------------------
    // prev_q/prev_u: previous optimal q (6 x N+1, column-major) and NORMALISED u (2 x N+1), used when solved.
    void linearization_nodes(const double q0[6], const double u0[3], double time_offset, bool solved,
                             const double* prev_q, const double* prev_u, Nodes& out) const {
        int Ns = TS.N_short, Nn = N() + 1;
        const std::vector<double>&ts = TS.ts, &dt = TS.dt, &prev_ts = TS.prev_ts;
        out.qs.assign(6 * Nn, 0); out.us.assign(2 * Nn, 0); out.ps.assign(4 * Nn, 0);
        double s0, e0, t0;
        traj.path_coordinates(q0[0], q0[1], s0, e0, t0);                      // :75
        TrajectoryNode tj = traj.at_s(s0);                                   // :76
        double ds = s0 - traj.at_time(ts[0]).s;                              // :77
        double dpsi = adiff(q0[2], tj.psi);                                  // :78
        double q[6] = {ds, q0[3], q0[4], q0[5], dpsi, e0};                   // :79
        double u[2] = {u0[0], u0[1] + u0[2]};                                // :80
        double p[4] = {tj.V, tj.kappa, 0, 0};                                // :81
        auto put = [&](int i) { for (int k = 0; k < 6; k++) out.qs[6 * i + k] = q[k]; out.us[2 * i] = u[0]; out.us[2 * i + 1] = u[1]; for (int k = 0; k < 4; k++) out.ps[4 * i + k] = p[k]; };
        if (solved) {                                                        // :82-102 (update_interpolations! :189-195 folded in)
            put(0);
            for (int i = 1; i < Nn; i++) {
                double t = ts[i];
                double tq = (t < prev_ts[Nn - 1]) ? t : prev_ts[Nn - 1];
                // Gridded(Linear()) on knots prev_ts (no extrapolation needed: tq <= last knot; below first knot -> weight < 0 never happens in closed loop,
                // Interpolations would throw a BoundsError; the restatement extends the first segment linearly)
                int j = TrajectoryTube::clampi(TrajectoryTube::searchsortedlast(prev_ts, tq), 1, Nn - 1) - 1;
                double w = (tq - prev_ts[j]) / (prev_ts[j + 1] - prev_ts[j]);
                for (int k = 0; k < 6; k++) q[k] = (1 - w) * prev_q[6 * j + k] + w * prev_q[6 * (j + 1) + k];
                for (int k = 0; k < 2; k++) u[k] = ((1 - w) * prev_u[2 * j + k] + w * prev_u[2 * (j + 1) + k]) * u_norm[k];
                double s = traj.at_time(t).s + q[0];                         // :96
                tj = traj.at_s(s);
                p[0] = tj.V; p[1] = tj.kappa; p[2] = 0; p[3] = 0;
                put(i);
            }
            return;
        }
        // cold start: :103-141
        double s = s0;
        double sdp = std::sin(dpsi), cdp = std::cos(dpsi);
        double V = q0[3] * cdp - q0[4] * sdp;
        double beta0 = std::atan2(q0[4], q0[3]);
        double r0 = q0[5], delta0 = u0[0];
        double Fyf0, Fyr0;
        lateral_tire_forces_q(veh, q0[3], q0[4], q0[5], u0[0], u0[1], u0[2], Fyf0, Fyr0);   // :110
        for (int i = 0; i < Nn; i++) {
            double tau = (i == Nn - 1) ? dt[i - 1] : dt[i];
            tj = traj.at_s(s);
            ds = s - traj.at_time(ts[i]).s;
            double A_des = tj.A + cp.k_V * (tj.V - V) / tau + (std::isnan(time_offset) ? 0.0 : -cp.k_s * ds / tau / tau);
            A_des = jl_min(jl_max(A_des, (cp.V_min - V) / tau), (cp.V_max - V) / tau);
            double A;
            if (i == 0) {
                double ud[2] = {u0[0], u0[1] + u0[2]}, qd[6];
                vehicle_world_dynamics<double>(veh, q0, ud, qd);             // :118 (LocalRoadGeometry only carries zero grade terms)
                A = (qd[3] - q0[5] * q0[4]) * cdp - (qd[4] + q0[5] * q0[3]) * sdp;
            } else if (i <= Ns) {
                SteadyState est = steady_state_estimates(veh, V, A_des, tj.kappa, 1, r0, beta0, delta0, Fyf0);
                q[0] = ds; q[1] = q0[3]; q[2] = q0[4]; q[3] = q0[5]; q[4] = adiff(q0[2], tj.psi); q[5] = e0;
                u[0] = est.delta; u[1] = est.Fxf + est.Fxr;
                p[0] = tj.V; p[1] = tj.kappa; p[2] = 0; p[3] = 0;
                A = est.A;
            } else {
                SteadyState est = steady_state_estimates(veh, V, A_des, tj.kappa, 4, V * tj.kappa, 0, 0, 0);
                q[0] = ds; q[1] = est.Ux; q[2] = est.Uy; q[3] = est.r; q[4] = -est.beta; q[5] = 0;
                u[0] = est.delta; u[1] = est.Fxf + est.Fxr;
                p[0] = tj.V; p[1] = tj.kappa; p[2] = 0; p[3] = 0;
                A = est.A;
            }
            put(i);
            if (i == Nn - 1) break;
            V = V + A * tau;
            s = s + V * tau + A * tau * tau / 2;
        }
    }

    // ---- propagate + linearize (third-party semantics restated, see header) ----------------------------------
    template <class T>
    void rk4_tracking(T q[6], const T u0[2], const T uf[2], const double p0[4], const double pf[4], double dt, bool ramp) const {
        int Nsub = rk4_substeps;
        double h = dt / Nsub;
        auto f = [&](const T x[6], double tau, T out[6]) {
            T u[2], p[4];
            if (ramp) {
                double w = tau / dt;
                for (int k = 0; k < 2; k++) u[k] = u0[k] + (uf[k] - u0[k]) * w;
                for (int k = 0; k < 4; k++) p[k] = T(p0[k] + (pf[k] - p0[k]) * w);
            } else {
                for (int k = 0; k < 2; k++) u[k] = u0[k];
                for (int k = 0; k < 4; k++) p[k] = T(p0[k]);
            }
            vehicle_tracking_dynamics<T>(veh, x, u, p, out);
        };
        for (int i = 0; i < Nsub; i++) {
            double t0 = i * h;
            T k1[6], k2[6], k3[6], k4[6], x[6];
            f(q, t0, k1);
            for (int k = 0; k < 6; k++) x[k] = q[k] + k1[k] * (h / 2);
            f(x, t0 + h / 2, k2);
            for (int k = 0; k < 6; k++) x[k] = q[k] + k2[k] * (h / 2);
            f(x, t0 + h / 2, k3);
            for (int k = 0; k < 6; k++) x[k] = q[k] + k3[k] * h;
            f(x, t0 + h, k4);
            for (int k = 0; k < 6; k++) q[k] = q[k] + (k1[k] + 2.0 * k2[k] + 2.0 * k3[k] + k4[k]) * (h / 6);
        }
    }
    // One interval: A (6x6 row-major), B0/Bf (6x2, NOT yet normalised), c (6).
    void linearize_interval(const double* q, const double* u0, const double* p0, const double* uf, const double* pf, double dt, bool ramp,
                            double* A, double* B0, double* Bf, double* c) const {
        typedef Dual<10> D;
        D x[6], du0[2], duf[2];
        for (int k = 0; k < 6; k++) x[k] = D::seed(q[k], k);
        for (int k = 0; k < 2; k++) { du0[k] = D::seed(u0[k], 6 + k); duf[k] = ramp ? D::seed(uf[k], 8 + k) : D(u0[k]); }
        rk4_tracking<D>(x, du0, duf, p0, ramp ? pf : p0, dt, ramp);
        for (int i = 0; i < 6; i++) {
            double ci = x[i].v;
            for (int j = 0; j < 6; j++) { A[6 * i + j] = x[i].d[j]; ci -= x[i].d[j] * q[j]; }
            for (int j = 0; j < 2; j++) {
                B0[2 * i + j] = x[i].d[6 + j]; ci -= x[i].d[6 + j] * u0[j];
                Bf[2 * i + j] = ramp ? x[i].d[8 + j] : 0.0;
                if (ramp) ci -= x[i].d[8 + j] * uf[j];
            }
            c[i] = ci;
        }
    }

    // ---- update_QP!: coupled_lat_long.jl:315-368 ---------------------------------------------------------------
    void update_qp(const Nodes& nd, const double current_state[6], const double current_control[3], const double other_car[4],
                   StageData& sd, double* V_hji_out = nullptr) const {
        int Ns = TS.N_short, Nl = TS.N_long, Nt = Ns + Nl;
        sd.resize(Ns, Nl);
        const double *qs = nd.qs.data(), *us = nd.us.data(), *ps = nd.ps.data();
        for (int k = 0; k < Nt; k++) sd.dt[k] = TS.dt[k];
        for (int k = 0; k < 6; k++) sd.q_curr[k] = qs[k];                    // :332
        for (int k = 0; k < 2; k++) sd.u_curr[k] = us[k] / u_norm[k];        // :333
        for (int t = 0; t < Nt; t++) {                                       // :335-340, :347-353
            bool ramp = t >= Ns;
            linearize_interval(qs + 6 * t, us + 2 * t, ps + 4 * t, us + 2 * (t + 1), ps + 4 * (t + 1), TS.dt[t], ramp,
                               &sd.A[36 * t], &sd.B0[12 * t], &sd.Bf[12 * t], &sd.c[6 * t]);
            for (int i = 0; i < 6; i++) for (int j = 0; j < 2; j++) { sd.B0[12 * t + 2 * i + j] *= u_norm[j]; sd.Bf[12 * t + 2 * i + j] *= u_norm[j]; }
        }
        // :341-346
        double x7[7], M[2], b, Vh;
        hji_relative_state(current_state, other_car, x7);
        double uR[2] = {current_control[0], current_control[1] + current_control[2]};
        reachability_constraint(veh, hji, x7, HJI_eps, uR, M, b, Vh);
        sd.M_hji[0] = M[0] * u_norm[0]; sd.M_hji[1] = M[1] * u_norm[1]; sd.b_hji = b;
        if (V_hji_out) *V_hji_out = Vh;
        // :354-367
        for (int t = 0; t < Nt; t++) {
            double Uxt = qs[6 * (t + 1) + 1];
            double Fxf, Fxr, Fx = us[2 * (t + 1) + 1];
            longitudinal_tire_forces<double>(veh, Fx, Fxf, Fxr);
            StableLimits sl = stable_limits(veh, Uxt, Fxf, Fxr);
            for (int i = 0; i < 4; i++) { sd.H[8 * t + 2 * i] = sl.H[i][0]; sd.H[8 * t + 2 * i + 1] = sl.H[i][1]; sd.G[4 * t + i] = sl.G[i]; }
            sd.dmin[t] = jl_max(sl.delta_min, -veh.delta_max) / u_norm[0];
            sd.dmax[t] = jl_min(sl.delta_max, veh.delta_max) / u_norm[0];
            sd.fxmax[t] = jl_min(veh.Px_max / Uxt, veh.Fx_max) / u_norm[1];
            sd.ddmin[t] = -cp.deltadot_max * TS.dt[t] / u_norm[0];
            sd.ddmax[t] = cp.deltadot_max * TS.dt[t] / u_norm[0];
        }
    }

    // get_next_control: coupled_lat_long.jl:370-374 (u2n = NORMALISED u[:,2] of the QP solution)
    void next_control(const double u2n[2], double out[3]) const {
        double d = u2n[0] * u_norm[0], Fx = u2n[1] * u_norm[1];
        double Fxf, Fxr;
        longitudinal_tire_forces<double>(veh, Fx, Fxf, Fxr);
        out[0] = d; out[1] = Fxf; out[2] = Fxr;
    }

    // simulate's plant step: model_predictive_control.jl:94  propagate(dynamics, state, StepControl(dt, BicycleControl2(u)))
    void plant_step(double q[6], const double u3[3], double dt) const {
        double u[2] = {u3[0], u3[1] + u3[2]};
        int Nsub = rk4_substeps; double h = dt / Nsub;
        for (int i = 0; i < Nsub; i++) {
            double k1[6], k2[6], k3[6], k4[6], x[6];
            vehicle_world_dynamics<double>(veh, q, u, k1);
            for (int k = 0; k < 6; k++) x[k] = q[k] + k1[k] * (h / 2);
            vehicle_world_dynamics<double>(veh, x, u, k2);
            for (int k = 0; k < 6; k++) x[k] = q[k] + k2[k] * (h / 2);
            vehicle_world_dynamics<double>(veh, x, u, k3);
            for (int k = 0; k < 6; k++) x[k] = q[k] + k3[k] * h;
            vehicle_world_dynamics<double>(veh, x, u, k4);
            for (int k = 0; k < 6; k++) q[k] += (k1[k] + 2 * k2[k] + 2 * k3[k] + k4[k]) * (h / 6);
        }
    }
};

}  // namespace po
