// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the shipped product path.
// CPU restatement of /root/reference/src/vehicles.jl and src/vehicle_dynamics.jl (line-by-line; each
// function cites the reference lines it follows).  Parity is UNPINNED: the reference holds no golden
// vectors for any of this (test/runtests.jl:1-5 is a placeholder).
#pragma once
#include <algorithm>
#include <cmath>
#include "dual.hpp"

namespace po {

// vehicles.jl:1-59 (X1 dictionary) flattened into one POD; vehicle_dynamics.jl:7-29,272-292 field sets.
struct VehicleParams {
    double G, m, Izz, L, a, b, h, mu, Caf, Car, Cd0, Cd1, Cd2;      // BicycleModelParams
    double fwd_frac, rwd_frac, fwb_frac, rwb_frac;                   // LongitudinalActuationParams
    double Fx_max, Fx_min, Px_max, delta_max, kappa_max;             // ControlLimits
};

// vehicles.jl:1-59
inline VehicleParams X1() {
    VehicleParams P;
    P.G = 9.80665;
    double mfl = 484, mfr = 455, mrl = 521, mrr = 504;
    P.m = mfl + mfr + mrl + mrr;
    P.Izz = 2900;
    P.L = 2.87;
    P.a = (mrl + mrr) / P.m * P.L;
    P.b = (mfl + mfr) / P.m * P.L;
    double hf = 0.1, hr = 0.1, h1 = 0.37;
    P.h = hf * P.b / P.L + hr * P.a / P.L + h1;
    P.mu = 0.92; P.Caf = 150e3; P.Car = 220e3;
    P.Fx_max = 5600; P.Px_max = 75e3;
    P.Cd0 = 241.0; P.Cd1 = 25.1; P.Cd2 = 0.0;
    P.fwd_frac = 0.0; P.rwd_frac = 1 - P.fwd_frac; P.fwb_frac = 0.6; P.rwb_frac = 1 - P.fwb_frac;
    P.Fx_min = std::max(-P.m * P.G * P.a * P.mu / (P.L * P.rwb_frac + P.mu * P.h),
                        -P.m * P.G * P.b * P.mu / (P.L * P.fwb_frac - P.mu * P.h));
    P.delta_max = 18 * M_PI / 180;
    P.kappa_max = std::tan(P.delta_max) / P.L;
    return P;
}

// Julia min/max propagate NaN; std::min/max and fmin/fmax do not.
inline double jl_min(double a, double b) { return (std::isnan(a) || std::isnan(b)) ? NAN : (b < a ? b : a); }
inline double jl_max(double a, double b) { return (std::isnan(a) || std::isnan(b)) ? NAN : (b > a ? b : a); }
inline double jl_clamp(double x, double lo, double hi) { return x > hi ? hi : (x < lo ? lo : x); }
template <int N> inline Dual<N> jl_clamp(const Dual<N>& x, double lo, double hi) { return x.v > hi ? Dual<N>(hi) : (x.v < lo ? Dual<N>(lo) : x); }

// vehicle_dynamics.jl:40-48
template <class T>
inline T _fialatiremodel(const T& tana, double Ca, const T& Fy_max) {
    T tana_slide = 3.0 * Fy_max / Ca;
    T ratio = absv(tana / tana_slide);
    if (ratio <= 1.0) return -Ca * tana * (1.0 - ratio + ratio * ratio / 3.0);
    return -Fy_max * signv(tana);
}
// vehicle_dynamics.jl:35-38
template <class T>
inline T fialatiremodel(const T& alpha, double Ca, double mu, const T& Fx, const T& Fz) {
    T F_max = mu * Fz;
    if (absv(Fx) >= F_max) return T(0.0);
    return _fialatiremodel(tan(alpha), Ca, sqrt(F_max * F_max - Fx * Fx));
}
// vehicle_dynamics.jl:56-62 (returns tan(alpha))
inline double _invfialatiremodel(double Fy, double Ca, double Fy_max) {
    if (std::fabs(Fy) >= Fy_max) return -(3 * Fy_max / Ca) * signv(Fy);
    return -(1 + std::cbrt(std::fabs(Fy) / Fy_max - 1)) * signv(Fy);
}

// vehicle_dynamics.jl:64-76
template <class T>
inline void lateral_tire_forces(const VehicleParams& B, const T& af, const T& ar, const T& Fxf, const T& Fxr,
                                const T& sd, const T& cd, T& Fyf, T& Fyr, int num_iters = 3) {
    Fyf = T(0.0);
    T Fx = Fxf * cd - Fyf * sd + Fxr;
    for (int i = 0; i < num_iters; i++) {
        T Fzf = (B.m * B.G * B.b - B.h * Fx) / B.L;
        Fyf = fialatiremodel(af, B.Caf, B.mu, Fxf, Fzf);
        Fx = Fxf * cd - Fyf * sd + Fxr;
    }
    T Fzr = (B.m * B.G * B.a + B.h * Fx) / B.L;
    Fyr = fialatiremodel(ar, B.Car, B.mu, Fxr, Fzr);
}
// vehicle_dynamics.jl:78-87 (state form; Ux,Uy,r are q[4..6] of a BicycleState)
inline void lateral_tire_forces_q(const VehicleParams& B, double Ux, double Uy, double r, double delta, double Fxf, double Fxr,
                                  double& Fyf, double& Fyr) {
    double sd = std::sin(delta), cd = std::cos(delta);
    double af = std::atan2(Uy + B.a * r, Ux) - delta;
    double ar = std::atan2(Uy - B.b * r, Ux);
    lateral_tire_forces<double>(B, af, ar, Fxf, Fxr, sd, cd, Fyf, Fyr);
}

// vehicle_dynamics.jl:279-283
template <class T>
inline void longitudinal_tire_forces(const VehicleParams& P, const T& Fx, T& Fxf, T& Fxr) {
    if (Fx > 0.0) { Fxf = Fx * P.fwd_frac; Fxr = Fx * P.rwd_frac; }
    else          { Fxf = Fx * P.fwb_frac; Fxr = Fx * P.rwb_frac; }
}
// vehicle_dynamics.jl:293-298; Ux enters by VALUE only (ForwardDiff.value, :295)
template <class T>
inline void apply_control_limits(const VehicleParams& P, const T& delta, const T& Fx, double Ux, T& delta_out, T& Fx_out) {
    delta_out = jl_clamp(delta, -P.delta_max, P.delta_max);
    // max(min(Fx, Fx_max, Px_max/Ux), Fx_min)
    double cap = jl_min(P.Fx_max, P.Px_max / Ux);
    T f = Fx;
    if (cap < value(f)) f = T(cap);
    if (P.Fx_min > value(f)) f = T(P.Fx_min);
    Fx_out = f;
}

// Shared body of BicycleModel / TrackingBicycleModel: vehicle_dynamics.jl:114-125 / :162-173
template <class T>
struct BodyForces { T dUx, dUy, dr; };
template <class T>
inline BodyForces<T> body_dynamics(const VehicleParams& B, const T& Ux, const T& Uy, const T& r, const T& delta, const T& Fxf, const T& Fxr) {
    T sd = sin(delta), cd = cos(delta);
    T af = atan2(Uy + B.a * r, Ux) - delta;
    T ar = atan2(Uy - B.b * r, Ux);
    T Fyf, Fyr;
    lateral_tire_forces<T>(B, af, ar, Fxf, Fxr, sd, cd, Fyf, Fyr);
    T Fx_drag = -B.Cd0 - Ux * (B.Cd1 + B.Cd2 * Ux);
    T Fxf_t = Fxf * cd - Fyf * sd;
    T Fyf_t = Fyf * cd + Fxf * sd;
    BodyForces<T> o;
    o.dUx = (Fxf_t + Fxr + Fx_drag) / B.m + r * Uy;
    o.dUy = (Fyf_t + Fyr) / B.m - r * Ux;
    o.dr = (B.a * Fyf_t - B.b * Fyr) / B.Izz;
    return o;
}

// VehicleModel{BicycleModel}: vehicle_dynamics.jl:310-313 wrapping :111-135.  q = (E,N,psi,Ux,Uy,r), u = (delta, Fx)
template <class T>
inline void vehicle_world_dynamics(const VehicleParams& P, const T q[6], const T u[2], T out[6]) {
    T d, Fx, Fxf, Fxr;
    apply_control_limits<T>(P, u[0], u[1], value(q[3]), d, Fx);
    longitudinal_tire_forces<T>(P, Fx, Fxf, Fxr);
    BodyForces<T> f = body_dynamics<T>(P, q[3], q[4], q[5], d, Fxf, Fxr);
    T sp = sin(q[2]), cp = cos(q[2]);
    out[0] = -q[3] * sp - q[4] * cp;     // psi measured from North (:127)
    out[1] = q[3] * cp - q[4] * sp;
    out[2] = q[5];
    out[3] = f.dUx; out[4] = f.dUy; out[5] = f.dr;
}

// VehicleModel{TrackingBicycleModel}: vehicle_dynamics.jl:310-315 wrapping :159-183.
// q = (ds,Ux,Uy,r,dpsi,e), u = (delta,Fx), p = (V,kappa,theta,phi)
template <class T>
inline void vehicle_tracking_dynamics(const VehicleParams& P, const T q[6], const T u[2], const T p[4], T out[6]) {
    T d, Fx, Fxf, Fxr;
    apply_control_limits<T>(P, u[0], u[1], value(q[1]), d, Fx);
    longitudinal_tire_forces<T>(P, Fx, Fxf, Fxr);
    BodyForces<T> f = body_dynamics<T>(P, q[1], q[2], q[3], d, Fxf, Fxr);
    T s = sin(q[4]), c = cos(q[4]);
    T vs = q[1] * c - q[2] * s;
    out[0] = vs - p[0];
    out[1] = f.dUx; out[2] = f.dUy; out[3] = f.dr;
    out[4] = q[3] - vs * p[1];
    out[5] = q[1] * s + q[2] * c;
}

// vehicle_dynamics.jl:227-263
struct StableLimits { double delta_min, delta_max, H[4][2], G[4]; };
inline StableLimits stable_limits(const VehicleParams& B, double Ux, double Fxf, double Fxr) {
    double Fx = Fxf + Fxr;
    double Fzf = (B.m * B.G * B.b - B.h * Fx) / B.L;
    double Fzr = (B.m * B.G * B.a + B.h * Fx) / B.L;
    double Ff_max = B.mu * Fzf, Fr_max = B.mu * Fzr;
    double Fyf_max = std::fabs(Fxf) > Ff_max ? 0.0 : std::sqrt(Ff_max * Ff_max - Fxf * Fxf);
    double Fyr_max = std::fabs(Fxr) > Fr_max ? 0.0 : std::sqrt(Fr_max * Fr_max - Fxr * Fxr);
    double tf = 3 * Fyf_max / B.Caf, tr = 3 * Fyr_max / B.Car;
    double af = std::atan(tf), ar = std::atan(tr);
    StableLimits o;
    o.delta_max = std::atan(B.L * (B.mu * B.G) / (Ux * Ux) - tr) + af;
    o.delta_min = std::atan(B.L * (-B.mu * B.G) / (Ux * Ux) + tr) - af;
    double rC = (B.mu * B.G) / Ux;
    double UyC = -Ux * tr + B.b * rC;
    double rD = Ux / B.L * (std::tan(af + o.delta_max) - tr);
    double UyD = Ux * tr + B.b * rD;
    double mCD = (rD - rC) / (UyD - UyC);
    double rE = Ux / B.L * (std::tan(-af + o.delta_min) + tr);
    double UyE = -Ux * tr + B.b * rE;
    double rF = (-B.mu * B.G) / Ux;
    double UyF = Ux * tr + B.b * rF;
    double mEF = (rF - rE) / (UyF - UyE);
    o.H[0][0] = 1 / Ux;  o.H[0][1] = -B.b / Ux;
    o.H[1][0] = -1 / Ux; o.H[1][1] = B.b / Ux;
    o.H[2][0] = -mCD;    o.H[2][1] = 1;
    o.H[3][0] = mEF;     o.H[3][1] = -1;
    o.G[0] = ar; o.G[1] = ar; o.G[2] = rC - UyC * mCD; o.G[3] = -rF + UyF * mEF;
    return o;
}

// vehicle_dynamics.jl:319-390
struct SteadyState { double beta, Ux, Uy, r, A, delta, Fxf, Fxr; };
inline SteadyState steady_state_estimates(const VehicleParams& P, double V, double A_tan, double kappa,
                                          int num_iters, double r, double beta0, double delta0, double Fyf0) {
    double A_rad = V * V * kappa;
    double A_mag = std::hypot(A_tan, A_rad);
    double A_max = P.mu * P.G;
    if (A_mag > A_max) {
        if (std::fabs(A_rad) > A_max) { A_rad = A_max * signv(A_rad); A_tan = 0.0; }
        else A_tan = std::sqrt(A_max * A_max - A_rad * A_rad) * signv(A_tan);
    }
    double rdot = A_tan * kappa;
    int i = 1;
    double beta = beta0, delta = delta0, Fyf = Fyf0;
    double Ux = 0, Uy = 0, Fxr = 0, Fxf = 0;
    while (true) {
        double sb = std::sin(beta), cb = std::cos(beta), sd = std::sin(delta), cd = std::cos(delta);
        Ux = V * cb; Uy = V * sb;
        double Fx_drag = -P.Cd0 - Ux * (P.Cd1 + P.Cd2 * Ux);
        double Ax = A_tan * cb - A_rad * sb;
        double Ay = A_tan * sb + A_rad * cb;
        double Fx = Ax * P.m - Fx_drag;
        Fx = jl_min(Fx, jl_min(P.Fx_max, P.Px_max / Ux) * (P.rwd_frac + P.fwd_frac * cd) - Fyf * sd);
        double Fzr = (P.m * P.G * P.a + P.h * Fx) / P.L, Fzf = (P.m * P.G * P.b - P.h * Fx) / P.L;
        double Fr_max = P.mu * Fzr, Ff_max = P.mu * Fzf;
        Fxr = jl_clamp((Fx + Fyf * sd) * (Fx > 0 ? P.rwd_frac / (P.rwd_frac + P.fwd_frac * cd)
                                                  : P.rwb_frac / (P.rwb_frac + P.fwb_frac * cd)), -Fr_max, Fr_max);
        double Fyr_max = std::sqrt(Fr_max * Fr_max - Fxr * Fxr);
        double Fyr = (Ay * P.m - rdot * P.Izz / P.a) / (1 + P.b / P.a);
        Fyr = jl_clamp(Fyr, -Fyr_max, Fyr_max);
        double tanar = _invfialatiremodel(Fyr, P.Car, Fyr_max);
        double Fxf_t = jl_clamp(Fx - Fxr, -Ff_max, Ff_max);
        double Fyf_tmax = std::sqrt(Ff_max * Ff_max - Fxf_t * Fxf_t);
        double Fyf_t = jl_clamp((P.b * Fyr + rdot * P.Izz) / P.a, -Fyf_tmax, Fyf_tmax);
        Fxf = Fxf_t * cd + Fyf_t * sd;
        Fyf = Fyf_t * cd - Fxf_t * sd;
        double Fyf_max = std::sqrt(Ff_max * Ff_max - Fxf * Fxf);
        double af = std::atan(_invfialatiremodel(Fyf, P.Caf, Fyf_max));
        delta = std::atan2(Uy + P.a * r, Ux) - af;
        if (i == num_iters) {
            Ax = (Fxf * cd - Fyf * sd + Fxr + Fx_drag) / P.m;    // NB: cd/sd are those of the delta the iteration started with (:378)
            Ay = (Fyf * cd + Fxf * sd + Fyr) / P.m;
            A_tan = Ax * cb + Ay * sb;
            break;
        }
        i++;
        beta = std::atan(tanar + P.b * r / Ux);
    }
    SteadyState o;
    o.beta = beta; o.Ux = V * std::cos(beta); o.Uy = V * std::sin(beta); o.r = r; o.A = A_tan; o.delta = delta; o.Fxf = Fxf; o.Fxr = Fxr;
    return o;
}

}  // namespace po
