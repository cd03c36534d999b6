// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the shipped product path.
// CPU restatement of /root/reference/src/HJI_computation.jl:1-131,160-170.  Parity UNPINNED: the
// reference's grid file (deps/BicycleCAvoid.jld2) is not in the repository (deps/build.jl:1-4) and no test
// pins a lookup.  7-D gridded multilinear interpolation restates Interpolations.jl 0.11.2 Gridded(Linear())
// (third-party, absent): per dim i = clamp(searchsortedlast(k,x),1,n-1), w = (x-k_i)/(k_{i+1}-k_i) with
// Float32 knots promoted to Float64, tensor-product weights, Float32 coefficients promoted to Float64.
#pragma once
#include <cmath>
#include <vector>
#include "dual.hpp"
#include "trajectory.hpp"
#include "vehicle.hpp"

namespace po {

struct HJICache {
    int dims[7] = {0, 0, 0, 0, 0, 0, 0};
    std::vector<float> knots[7];
    std::vector<float> V;        // column-major, dim 1 fastest (Julia Array{Float32,7})
    std::vector<float> gradV;    // 7 floats per node (SVector{7,Float32} AoS), same node order
    bool loaded = false;

    // HJI_computation.jl:32-37
    void placeholder() {
        size_t n = 1;
        for (int d = 0; d < 7; d++) { dims[d] = 2; knots[d] = {-1000.f, 1000.f}; n *= 2; }
        V.assign(n, 0.f); gradV.assign(7 * n, 0.f); loaded = true;
    }
    static int searchsortedlast_f(const std::vector<float>& v, double x) {
        int lo = 0, hi = (int)v.size();
        while (lo < hi) { int mid = (lo + hi) / 2; if ((double)v[mid] <= x) lo = mid + 1; else hi = mid; }
        return lo;
    }
    // getindex: HJI_computation.jl:66-72.  Returns false (V=Inf, gradV=0) when out of bounds.
    bool lookup(const double x[7], double& Vout, double g[7]) const {
        // Build-defined: with NO grid installed the constraint is inactive (V = Inf), i.e. BASELINE config 2 ("HJI inactive").
        // The reference's own default, placeholder_HJICache() (:32-37), combined with a zero other-car state yields b_HJI = NaN
        // (SURVEY.md H4); call placeholder() explicitly to reproduce that.
        bool inb = loaded;
        for (int d = 0; d < 7 && inb; d++) inb = inb && ((double)knots[d][0] <= x[d]) && (x[d] <= (double)knots[d][dims[d] - 1]);
        if (!inb) { Vout = INFINITY; for (int k = 0; k < 7; k++) g[k] = 0; return false; }
        int idx[7]; double w[7]; size_t stride[7]; size_t st = 1;
        for (int d = 0; d < 7; d++) {
            int i = searchsortedlast_f(knots[d], x[d]);
            i = i < 1 ? 1 : (i > dims[d] - 1 ? dims[d] - 1 : i);
            idx[d] = i - 1;
            double k0 = knots[d][i - 1], k1 = knots[d][i];
            w[d] = (x[d] - k0) / (k1 - k0);
            stride[d] = st; st *= dims[d];
        }
        double accV = 0, accG[7] = {0, 0, 0, 0, 0, 0, 0};
        for (int c = 0; c < 128; c++) {
            double wt = 1; size_t off = 0;
            for (int d = 0; d < 7; d++) {
                int bit = (c >> d) & 1;
                wt *= bit ? w[d] : (1 - w[d]);
                off += (size_t)(idx[d] + bit) * stride[d];
            }
            accV += wt * (double)V[off];
            for (int k = 0; k < 7; k++) accG[k] += wt * (double)gradV[7 * off + k];
        }
        Vout = accV; for (int k = 0; k < 7; k++) g[k] = accG[k];
        return true;
    }
};

// HJIRelativeState(us, them): HJI_computation.jl:20-24.  NB `cpsi, spsi = sincos(-psi)` binds cpsi=sin(-psi), spsi=cos(-psi).
inline void hji_relative_state(const double us[6], const double them[4], double x[7]) {
    double cpsi = std::sin(-us[2]), spsi = std::cos(-us[2]);
    double dE = them[0] - us[0], dN = them[1] - us[1];
    x[0] = cpsi * dE + spsi * dN;
    x[1] = -spsi * dE + cpsi * dN;
    x[2] = adiff(them[2], us[2]);
    x[3] = us[3]; x[4] = us[4]; x[5] = them[3]; x[6] = us[5];
}

// relative_dynamics: HJI_computation.jl:74-88
template <class T>
inline void relative_dynamics(const VehicleParams& P, const double x[7], const T uR[2], const double uH[2], T out[7]) {
    T q[6] = {T(x[0]), T(x[1]), T(x[2]), T(x[3]), T(x[4]), T(x[6])};
    T bd[6];
    vehicle_world_dynamics<T>(P, q, uR, bd);
    double s = std::sin(x[2]), c = std::cos(x[2]);
    out[0] = T(x[5] * c - x[3] + x[1] * x[6]);
    out[1] = T(x[5] * s - x[4] - x[0] * x[6]);
    out[2] = T(uH[0] - x[6]);
    out[3] = bd[3]; out[4] = bd[4]; out[5] = T(uH[1]); out[6] = bd[5];
}

// optimal_disturbance (dMode = :min): HJI_computation.jl:90-131.  NaN behaviour of the reference is kept
// (V = 0 with an in-grid state gives lam_Ay = NaN; every comparison with NaN is false).
inline void optimal_disturbance(const VehicleParams& P, const double x[7], const double g[7], double uH[2]) {
    double Ax_max = P.Fx_max / P.m, Pmx_max = P.Px_max / P.m, maxA = 0.9 * P.mu * P.G;
    double sgn = -1;
    double V = x[5];
    double lam_w = g[2], lam_Ax = g[5];
    double lam_Ay = lam_w / V;
    double lam_norm = std::hypot(lam_Ax, lam_Ay);
    if (std::isnan(lam_Ax) || std::isnan(lam_Ay)) lam_norm = NAN;    // Julia hypot(NaN, x) is NaN unless x is Inf
    if (lam_norm < 1e-3) { uH[0] = 0; uH[1] = 0; return; }
    double desAx = sgn * lam_Ax * maxA / lam_norm, desAy = sgn * lam_Ay * maxA / lam_norm;
    double maxAx = jl_min(Ax_max, Pmx_max / V);
    double maxAy = P.kappa_max * V * V;
    if (desAx > maxAx) {
        if (std::fabs(desAy) < maxAy) maxAy = jl_min(maxAy, std::sqrt(maxA * maxA - maxAx * maxAx));
        uH[0] = std::copysign(maxAy, desAy) / V; uH[1] = maxAx; return;
    }
    if (std::fabs(desAy) > maxAy) {
        if (desAx > 0) { maxAx = jl_min(std::sqrt(maxA * maxA - maxAy * maxAy), maxAx); uH[0] = std::copysign(maxAy, desAy) / V; uH[1] = maxAx; return; }
        uH[0] = std::copysign(maxAy, desAy) / V; uH[1] = -std::sqrt(maxA * maxA - maxAy * maxAy); return;
    }
    uH[0] = desAy / V; uH[1] = maxAx;
}

// optimal_control (uMode = :max, N = 50): HJI_computation.jl:133-158.  Bang-bang steer from the sign of the lateral co-state, then a 50-point
// line search over Fx in [Fx_min, Fx_max] of grad V . (forces); strict '>' keeps the FIRST maximiser; a NaN score never wins (Fx_opt stays 0).
inline void optimal_control(const VehicleParams& P, const double x[7], const double g[7], double u2[2], int N = 50) {
    double A = g[3] / P.m, B = g[4] / P.m + P.a * g[6] / P.Izz, C = g[4] / P.m - P.b * g[6] / P.Izz;
    double d_opt = B >= 0 ? P.delta_max : -P.delta_max;
    double V_opt = -INFINITY, Fx_opt = 0.0;
    for (int n = 0; n < N; n++) {
        double frac = (double)n / (double)(N - 1);
        double Fx = frac * P.Fx_max + (1 - frac) * P.Fx_min;
        double Fxf, Fxr, Fyf, Fyr;
        longitudinal_tire_forces<double>(P, Fx, Fxf, Fxr);
        lateral_tire_forces_q(P, x[3], x[4], x[6], d_opt, Fxf, Fxr, Fyf, Fyr);      // fake_qR = (0,0,0,Ux,Uy,r)  (:137)
        double V = A * Fx + B * Fyf + C * Fyr;
        if (V > V_opt) { Fx_opt = Fx; V_opt = V; }
    }
    u2[0] = d_opt; u2[1] = Fx_opt;
}

// compute_reachability_constraint: HJI_computation.jl:160-170 with uR_lin = BicycleControl2(current_control)
// as passed at coupled_lat_long.jl:342.  Returns M (2), b, and the looked-up V.
inline void reachability_constraint(const VehicleParams& P, const HJICache& cache, const double x[7], double eps,
                                    const double uR_lin[2], double M[2], double& b, double& Vout) {
    double g[7];
    cache.lookup(x, Vout, g);
    if (Vout > eps) { M[0] = 0; M[1] = 0; b = 1.0; return; }
    double uH[2];
    optimal_disturbance(P, x, g, uH);
    Dual<2> uR[2] = {Dual<2>::seed(uR_lin[0], 0), Dual<2>::seed(uR_lin[1], 1)};
    Dual<2> f[7];
    relative_dynamics<Dual<2>>(P, x, uR, uH, f);
    Dual<2> Hm(0.0);
    for (int k = 0; k < 7; k++) Hm = Hm + f[k] * g[k];
    M[0] = Hm.d[0]; M[1] = Hm.d[1];
    b = Hm.v - (M[0] * uR_lin[0] + M[1] * uR_lin[1]);
}

}  // namespace po
