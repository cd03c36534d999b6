// Device-side vehicle physics and trajectory lookups for the MI355X MPC hot path (gfx950, fp64).
// Written from the reference's equations (file:line cited per function, relative to /root/reference/src);
// independent of oracle/ (the oracle is test infrastructure and is never included here).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/pigeon_mpc.h"

#define PG_DEV __device__ __forceinline__

namespace pg {

// Precision of the path arithmetic.  Every kernel is written against the scalar type `real`: one translation unit per instantiation, real = double
// for libpigeon_hip.so (the reference's type) and real = float (-DPG_F32) for libpigeon_hip_f32.so (BASELINE configs 3/4).  Floating literals are
// spelled real(...) so that no expression is silently promoted to double in the fp32 build; tools/check_f32_purity.py disassembles the fp32 code
// object and fails the build when an fp64 arithmetic instruction shows up outside the kernels that own absolute time.  `tdouble` marks the few
// quantities that stay fp64 in both builds: absolute time (the time grid's ceil() lattice must not depend on the arithmetic type of the QP).
#ifdef PG_F32
typedef float real; typedef float2 real2;
#define PG_BIG 1e30f
#else
typedef double real; typedef double2 real2;
#define PG_BIG 1e300
#endif
typedef double tdouble;
// device-side mirrors of pg_vehicle / pg_control_params (include/pigeon_mpc.h) in the arithmetic type of the build
struct DevVehicle { real G, m, Izz, L, a, b, h, mu, Caf, Car, Cd0, Cd1, Cd2, fwd_frac, rwd_frac, fwb_frac, rwb_frac, Fx_max, Fx_min, Px_max, delta_max, kappa_max; };
struct DevControl { real V_min, V_max, k_V, k_s, deltadot_max, Q_ds, Q_dpsi, Q_e, W_beta, W_r, W_HJI, R_delta, R_ddelta, R_Fx, R_dFx; int N_HJI; };
// reciprocal: hardware seed + Newton steps (~1 ulp); the IEEE division sequence costs ~5x more issue slots
#ifdef PG_F32
PG_DEV float frcp(float x) { float r = __builtin_amdgcn_rcpf(x); float e = fmaf(-x, r, 1.0f); return fmaf(r, e, r); }
PG_DEV void pg_sincos(float x, float* s, float* c) { sincosf(x, s, c); }
PG_DEV float pg_rsqrt(float x) { return rsqrtf(x); }
#else
PG_DEV double frcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    double e = fma(-x, r, 1.0); r = fma(r, e, r);
    e = fma(-x, r, 1.0); r = fma(r, e, r);
    return r;
}
// sin and cos together.  |x| <= pi/4 (every steering angle: |delta| <= 0.314; heading errors in normal operation) takes the two kernel polynomials of the
// argument-reduced range directly (Cody-Waite / fdlibm __kernel_sin, __kernel_cos coefficients, < 1 ulp) -- 15 FMAs instead of the library's ~100
// instructions of range reduction and selection; k_linearize evaluates 80 of these per lane.  Larger arguments go to the library (wave-level branch).
PG_DEV void pg_sincos(double x, double* s, double* c) {
    if (__builtin_expect(fabs(x) > 0.78539816339744830962, 0)) { sincos(x, s, c); return; }
    const double z = x * x;
    const double rs = fma(z, fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08), 2.75573137070700676789e-06), -1.98412698298579493134e-04),
                                 8.33333333332248946124e-03), -1.66666666666666324348e-01);
    *s = fma(x * z, rs, x);
    const double rc = fma(z, fma(z, fma(z, fma(z, fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09), -2.75573143513906633035e-07), 2.48015872894767294178e-05),
                                 -1.38888888888741095749e-03), 4.16666666666666019037e-02);
    *c = fma(z * z, rc, fma(z, -0.5, 1.0));
}
PG_DEV double pg_rsqrt(double x) { return rsqrt(x); }
#endif

// ---- two-tangent forward-mode number (stands in for ForwardDiff.Dual inside `linearize` and HJI_computation.jl:167) ----
struct D2 {
    real v, a, b;
    PG_DEV D2() {}
    PG_DEV D2(real x) : v(x), a(real(0.0)), b(real(0.0)) {}
    PG_DEV D2(real x, real da, real db) : v(x), a(da), b(db) {}
};
PG_DEV D2 operator+(D2 x, D2 y) { return D2(x.v + y.v, x.a + y.a, x.b + y.b); }
PG_DEV D2 operator-(D2 x, D2 y) { return D2(x.v - y.v, x.a - y.a, x.b - y.b); }
PG_DEV D2 operator-(D2 x) { return D2(-x.v, -x.a, -x.b); }
PG_DEV D2 operator*(D2 x, D2 y) { return D2(x.v * y.v, x.a * y.v + x.v * y.a, x.b * y.v + x.v * y.b); }
PG_DEV D2 operator/(D2 x, D2 y) { real inv = frcp(y.v), q = x.v * inv; return D2(q, (x.a - q * y.a) * inv, (x.b - q * y.b) * inv); }
PG_DEV D2 operator+(D2 x, real y) { return D2(x.v + y, x.a, x.b); }
PG_DEV D2 operator+(real y, D2 x) { return D2(x.v + y, x.a, x.b); }
PG_DEV D2 operator-(D2 x, real y) { return D2(x.v - y, x.a, x.b); }
PG_DEV D2 operator-(real y, D2 x) { return D2(y - x.v, -x.a, -x.b); }
PG_DEV D2 operator*(D2 x, real y) { return D2(x.v * y, x.a * y, x.b * y); }
PG_DEV D2 operator*(real y, D2 x) { return D2(x.v * y, x.a * y, x.b * y); }
PG_DEV D2 operator/(D2 x, real y) { real inv = frcp(y); return D2(x.v * inv, x.a * inv, x.b * inv); }
PG_DEV D2 operator/(real x, D2 y) { real inv = frcp(y.v), q = x * inv; return D2(q, -q * y.a * inv, -q * y.b * inv); }

PG_DEV real val(real x) { return x; }
PG_DEV real val(D2 x) { return x.v; }
PG_DEV D2 chain(D2 x, real f, real df) { return D2(f, df * x.a, df * x.b); }

PG_DEV void sincos_(real x, real& s, real& c) { pg_sincos(x, &s, &c); }
PG_DEV void sincos_(D2 x, D2& s, D2& c) { real sv, cv; pg_sincos(x.v, &sv, &cv); s = chain(x, sv, cv); c = chain(x, cv, -sv); }
PG_DEV real tan_(real x) { return tan(x); }
PG_DEV D2 tan_(D2 x) { real t = tan(x.v); return chain(x, t, real(1.0) + t * t); }
PG_DEV real sqrt_(real x) { return sqrt(x); }
PG_DEV D2 sqrt_(D2 x) { real s = sqrt(x.v); return chain(x, s, real(0.5) * frcp(s)); }
PG_DEV real atan2_(real y, real x) { return atan2(y, x); }
PG_DEV D2 atan2_(D2 y, D2 x) { real inv = real(1.0) / (x.v * x.v + y.v * y.v); return D2(atan2(y.v, x.v), (x.v * y.a - y.v * x.a) * inv, (x.v * y.b - y.v * x.b) * inv); }
PG_DEV real abs_(real x) { return fabs(x); }
PG_DEV D2 abs_(D2 x) { return x.v < real(0.0) ? -x : x; }
PG_DEV real sgn(real x) { return (real)((x > real(0.0)) - (x < real(0.0))); }
template <class T> PG_DEV T cst(real x);
template <> PG_DEV real cst<real>(real x) { return x; }
template <> PG_DEV D2 cst<D2>(real x) { return D2(x); }

// ---- K-tangent forward-mode number (k_linearize carries 4 directions per lane: one primal trajectory serves four Jacobian columns) ----
template <int K> struct DK {
    real v, d[K];
    PG_DEV DK() {}
    PG_DEV DK(real x) : v(x) {
#pragma unroll
        for (int k = 0; k < K; k++) d[k] = real(0.0);
    }
};
#define PG_DK_LOOP _Pragma("unroll") for (int k = 0; k < K; k++)
#define PG_DK_LOOP_T(T) _Pragma("unroll") for (int k = 0; k < (int)(sizeof(T::d) / sizeof(real)); k++)
template <int K> PG_DEV DK<K> operator+(DK<K> x, DK<K> y) { DK<K> r; r.v = x.v + y.v; PG_DK_LOOP r.d[k] = x.d[k] + y.d[k]; return r; }
template <int K> PG_DEV DK<K> operator-(DK<K> x, DK<K> y) { DK<K> r; r.v = x.v - y.v; PG_DK_LOOP r.d[k] = x.d[k] - y.d[k]; return r; }
template <int K> PG_DEV DK<K> operator-(DK<K> x) { DK<K> r; r.v = -x.v; PG_DK_LOOP r.d[k] = -x.d[k]; return r; }
template <int K> PG_DEV DK<K> operator*(DK<K> x, DK<K> y) { DK<K> r; r.v = x.v * y.v; PG_DK_LOOP r.d[k] = x.d[k] * y.v + x.v * y.d[k]; return r; }
template <int K> PG_DEV DK<K> operator/(DK<K> x, DK<K> y) { real inv = frcp(y.v), q = x.v * inv; DK<K> r; r.v = q; PG_DK_LOOP r.d[k] = (x.d[k] - q * y.d[k]) * inv; return r; }
template <int K> PG_DEV DK<K> operator+(DK<K> x, real y) { x.v += y; return x; }
template <int K> PG_DEV DK<K> operator+(real y, DK<K> x) { x.v += y; return x; }
template <int K> PG_DEV DK<K> operator-(DK<K> x, real y) { x.v -= y; return x; }
template <int K> PG_DEV DK<K> operator-(real y, DK<K> x) { DK<K> r; r.v = y - x.v; PG_DK_LOOP r.d[k] = -x.d[k]; return r; }
template <int K> PG_DEV DK<K> operator*(DK<K> x, real y) { x.v *= y; PG_DK_LOOP x.d[k] *= y; return x; }
template <int K> PG_DEV DK<K> operator*(real y, DK<K> x) { x.v *= y; PG_DK_LOOP x.d[k] *= y; return x; }
template <int K> PG_DEV DK<K> operator/(DK<K> x, real y) { real inv = frcp(y); x.v *= inv; PG_DK_LOOP x.d[k] *= inv; return x; }
template <int K> PG_DEV DK<K> operator/(real x, DK<K> y) { real inv = frcp(y.v), q = x * inv; DK<K> r; r.v = q; PG_DK_LOOP r.d[k] = -q * y.d[k] * inv; return r; }
template <int K> PG_DEV real val(DK<K> x) { return x.v; }
template <int K> PG_DEV DK<K> chain(DK<K> x, real f, real df) { DK<K> r; r.v = f; PG_DK_LOOP r.d[k] = df * x.d[k]; return r; }
template <int K> PG_DEV void sincos_(DK<K> x, DK<K>& s, DK<K>& c) { real sv, cv; pg_sincos(x.v, &sv, &cv); s = chain(x, sv, cv); c = chain(x, cv, -sv); }
template <int K> PG_DEV DK<K> sqrt_(DK<K> x) { real s = sqrt(x.v); return chain(x, s, real(0.5) * frcp(s)); }
template <int K> PG_DEV DK<K> abs_(DK<K> x) { return x.v < real(0.0) ? -x : x; }
template <> PG_DEV DK<4> cst<DK<4>>(real x) { return DK<4>(x); }
template <> PG_DEV DK<3> cst<DK<3>>(real x) { return DK<3>(x); }
template <> PG_DEV DK<2> cst<DK<2>>(real x) { return DK<2>(x); }
template <> PG_DEV DK<1> cst<DK<1>>(real x) { return DK<1>(x); }

// Julia min/max propagate NaN
PG_DEV real jmin(real a, real b) { return (a != a || b != b) ? NAN : (b < a ? b : a); }
PG_DEV real jmax(real a, real b) { return (a != a || b != b) ? NAN : (b > a ? b : a); }
PG_DEV real clampd(real x, real lo, real hi) { return x > hi ? hi : (x < lo ? lo : x); }

// ---- tire model: vehicle_dynamics.jl:35-48 ----
// takes tan(alpha) directly: the slip angles of the reference are atan(...) - delta, and only their tangent is ever used
// (vehicle_dynamics.jl:37), so tan(atan(y/x) - delta) = (y/x - tan delta)/(1 + (y/x) tan delta) replaces an atan2 + tan pair
// (identical in exact arithmetic for Ux > 0, the only regime the MPC runs in: V_min = 1, ros_integration.jl:84-87)
template <class T> struct is_dk { static constexpr bool value = false; };
template <int K> struct is_dk<DK<K>> { static constexpr bool value = true; };
// 1 / sqrt(x) with tangents: d(x^-1/2) = -1/2 x^-3/2 dx
template <int K> PG_DEV DK<K> rsqrt_(DK<K> x) { real z = pg_rsqrt(x.v); return chain(x, z, real(-0.5) * z * z * z); }
template <class T>
PG_DEV T fiala(T tana, real Ca, real mu, T Fx, T Fz) {
    T Fmax = mu * Fz;
    if (abs_(val(Fx)) >= val(Fmax)) return cst<T>(real(0.0));
    if constexpr (is_dk<T>::value) {
        // forward-mode numbers (k_linearize: four of these per dynamics evaluation, half of its instructions): the tire force is treated as ONE elementary function
        // of three scalars, Fy(tan alpha, Fx, Fz) -- value and the three partial derivatives in scalar arithmetic, then one three-term combination per direction --
        // instead of carrying every direction through each of its ~20 operations.  With z = 1 / sqrt(w), w = (mu Fz)^2 - Fx^2, rho = |tan alpha| z Ca / 3:
        //   adhesion (rho <= 1):  Fy = -Ca tan(alpha) (1 - rho + rho^2/3),   dFy/dtan = -Ca (1 - rho)^2,   dFy/dw = Ca tan(alpha) (2 rho / 3 - 1) rho z^2 / 2
        //   sliding:              Fy = -sqrt(w) sgn,                         dFy/dtan = 0,                 dFy/dw = -sgn z / 2
        // and dw = 2 mu^2 Fz dFz - 2 Fx dFx.  ONE reciprocal square root serves the value and the derivatives.  Same function as the generic branch below, other rounding (1e-16).
        const real fm = val(Fmax), fx = val(Fx), ta = val(tana);
        const real w = fm * fm - fx * fx, z = pg_rsqrt(w);
        const real rho = fabs(ta * z) * (Ca * (real(1.0) / real(3.0)));
        real Fy, g_t, g_w;
        if (rho <= real(1.0)) {
            const real cat = Ca * ta, omr = real(1.0) - rho;
            Fy = -cat * (omr + rho * rho * (real(1.0) / real(3.0)));
            g_t = -Ca * omr * omr;
            g_w = cat * (rho * (real(2.0) / real(3.0)) - real(1.0)) * rho * (real(0.5) * z * z);
        } else {
            const real sg = sgn(ta);
            Fy = -(w * z) * sg; g_t = real(0.0); g_w = real(-0.5) * z * sg;
        }
        const real g_z = g_w * (real(2.0) * fm * mu), g_x = g_w * (real(-2.0) * fx);
        T r; r.v = Fy;
        PG_DK_LOOP_T(T) r.d[k] = g_t * tana.d[k] + g_z * Fz.d[k] + g_x * Fx.d[k];
        return r;
    }
    T Fy_max = sqrt_(Fmax * Fmax - Fx * Fx);
    T slide = (real(3.0) / Ca) * Fy_max;
    T ratio = abs_(tana / slide);
    if (val(ratio) <= real(1.0)) return -(Ca * tana) * (real(1.0) - ratio + ratio * ratio * (real(1.0) / real(3.0)));
    return -Fy_max * sgn(val(tana));
}
// vehicle_dynamics.jl:56-62 (returns tan(alpha))
PG_DEV real inv_fiala_tan(real Fy, real Ca, real Fy_max) {
    if (fabs(Fy) >= Fy_max) return -(real(3.0) * Fy_max / Ca) * sgn(Fy);
    return -(real(1.0) + cbrt(fabs(Fy) / Fy_max - real(1.0))) * sgn(Fy);
}
// same with 3 / Ca precomputed
PG_DEV real inv_fiala_tan3(real Fy, real three_over_Ca, real Fy_max) {
    if (fabs(Fy) >= Fy_max) return -(Fy_max * three_over_Ca) * sgn(Fy);
    return -(real(1.0) + cbrt(fabs(Fy) / Fy_max - real(1.0))) * sgn(Fy);
}
// vehicle_dynamics.jl:64-76: 3-iteration front-axle load-transfer fixed point, then rear
template <class T>
PG_DEV void lateral_forces(const DevVehicle& P, T af, T ar, T Fxf, T Fxr, T sd, T cd, T& Fyf, T& Fyr, T* Fxf_t_out = nullptr) {   // af, ar: TANGENTS of the slip angles
    const real W_b = P.m * P.G * P.b, W_a = P.m * P.G * P.a, invL = real(1.0) / P.L;
    const real hL = P.h * invL, WbL = W_b * invL, WaL = W_a * invL;        // (forward-mode numbers: one multiplication per direction instead of two)
    Fyf = cst<T>(real(0.0));
    const T FxfC = Fxf * cd;                       // (formed once: the loop and the caller used to multiply it out four times; same association, same bits)
    T Fxt = FxfC;
    T Fx = FxfC + Fxr;
#pragma unroll 1
    for (int i = 0; i < 3; i++) {
        T Fzf;
        if constexpr (is_dk<T>::value) Fzf = WbL - hL * Fx; else Fzf = (W_b - P.h * Fx) * invL;
        Fyf = fiala<T>(af, P.Caf, P.mu, Fxf, Fzf);
        Fxt = FxfC - Fyf * sd;                     // the longitudinal front force in the body frame at this load-transfer iterate
        Fx = Fxt + Fxr;
    }
    if (Fxf_t_out) *Fxf_t_out = Fxt;
    T Fzr;
    if constexpr (is_dk<T>::value) Fzr = WaL + hL * Fx; else Fzr = (W_a + P.h * Fx) * invL;
    Fyr = fiala<T>(ar, P.Car, P.mu, Fxr, Fzr);
}

// apply_control_limits (vehicle_dynamics.jl:293-298; Ux by value) followed by longitudinal_tire_forces (:279-283)
template <class T>
PG_DEV void actuate(const DevVehicle& P, T delta, T Fx, real Ux, T& d_out, T& Fxf, T& Fxr) {
    real dv = val(delta);
    d_out = dv > P.delta_max ? cst<T>(P.delta_max) : (dv < -P.delta_max ? cst<T>(-P.delta_max) : delta);
    real cap = jmin(P.Fx_max, P.Px_max / Ux);
    T f = Fx;
    if (cap < val(f)) f = cst<T>(cap);
    if (P.Fx_min > val(f)) f = cst<T>(P.Fx_min);
    if (val(f) > real(0.0)) { Fxf = f * P.fwd_frac; Fxr = f * P.rwd_frac; }
    else              { Fxf = f * P.fwb_frac; Fxr = f * P.rwb_frac; }
}

// body-frame accelerations common to BicycleModel (:114-133) and TrackingBicycleModel (:162-178)
template <class T>
PG_DEV void body_accel(const DevVehicle& P, T Ux, T Uy, T r, T delta, T Fxf, T Fxr, T& dUx, T& dUy, T& dr) {
    T sd, cd; sincos_(delta, sd, cd);
    T taf, tar, Fx_drag;
    if constexpr (is_dk<T>::value) {
        // forward-mode numbers: tan(alpha_f) as ONE elementary function of (y = Uy + a r, Ux, delta) -- value and three partial derivatives in scalars, one
        // three-term combination per direction -- instead of carrying every direction through two quotients and the tangent-difference formula; likewise the drag
        const real iux = frcp(val(Ux)), yv = val(Uy) + P.a * val(r), tfv = yv * iux, tdv = val(sd) * frcp(val(cd));
        const real iden = frcp(real(1.0) + tfv * tdv), id2 = iden * iden;
        const real c_y = (real(1.0) + tdv * tdv) * id2 * iux, c_u = -c_y * tfv, c_d = -(real(1.0) + tfv * tfv) * id2 * (real(1.0) + tdv * tdv);
        taf.v = (tfv - tdv) * iden;
        const real dd = -(P.Cd1 + real(2.0) * P.Cd2 * val(Ux));
        Fx_drag.v = -P.Cd0 - val(Ux) * (P.Cd1 + P.Cd2 * val(Ux));
        PG_DK_LOOP_T(T) { taf.d[k] = c_y * (Uy.d[k] + P.a * r.d[k]) + c_u * Ux.d[k] + c_d * delta.d[k]; Fx_drag.d[k] = dd * Ux.d[k]; }
        tar.v = (val(Uy) - P.b * val(r)) * iux;                                   // tan(atan2(Uy - b r, Ux)) with the reciprocal at hand
        const real t_u = -tar.v * iux;
        PG_DK_LOOP_T(T) tar.d[k] = iux * (Uy.d[k] - P.b * r.d[k]) + t_u * Ux.d[k];
    } else {
        T tf = (Uy + P.a * r) / Ux, td = sd / cd;
        taf = (tf - td) / (real(1.0) + tf * td);          // tan(atan2(Uy + a r, Ux) - delta)   (:118)
        Fx_drag = -P.Cd0 - Ux * (P.Cd1 + P.Cd2 * Ux);
        tar = (Uy - P.b * r) / Ux;                 // tan(atan2(Uy - b r, Ux))           (:119)
    }
    T Fyf, Fyr;
    T Fxf_t;
    lateral_forces<T>(P, taf, tar, Fxf, Fxr, sd, cd, Fyf, Fyr, &Fxf_t);
    T Fyf_t = Fyf * cd + Fxf * sd;
    const real invm = real(1.0) / P.m, invI = real(1.0) / P.Izz;
    dUx = (Fxf_t + Fxr + Fx_drag) * invm + r * Uy;
    dUy = (Fyf_t + Fyr) * invm - r * Ux;
    dr = (P.a * Fyf_t - P.b * Fyr) * invI;
}

// VehicleModel{TrackingBicycleModel}: vehicle_dynamics.jl:310-315 over :159-183.  q=(ds,Ux,Uy,r,dpsi,e), u=(delta,Fx), p=(V,kappa)
template <class T>
PG_DEV void tracking_rhs(const DevVehicle& P, const T q[6], T u0, T u1, real pV, real pK, T out[6]) {
    T d, Fxf, Fxr;
    actuate<T>(P, u0, u1, val(q[1]), d, Fxf, Fxr);
    T s, c; sincos_(q[4], s, c);
    T vs = q[1] * c - q[2] * s;
    out[0] = vs - pV;
    body_accel<T>(P, q[1], q[2], q[3], d, Fxf, Fxr, out[1], out[2], out[3]);
    out[4] = q[3] - vs * pK;
    out[5] = q[1] * s + q[2] * c;
}
// ---- the same right-hand side with its LOCAL JACOBIAN in scalar arithmetic (k_linearize since round 4) ----
// `linearize` needs d(flow map)/d(q, u0, uf): 8 tangent directions through 40 evaluations of the dynamics per interval.  Rounds 1-3 carried every direction through
// every operation of the right-hand side (forward-mode numbers DK<K>, ~120 fp64 operations per direction and evaluation, and 5-vectors for every intermediate: the
// kernel ran at 360 registers).  The right-hand side has only FIVE inputs that matter to its three non-trivial outputs -- (Ux, Uy, r, delta, Fx) -> (Ux', Uy', r') --
// and the other three outputs are kinematics of (Ux, Uy, dpsi, r).  So: value and the 3 x 5 + 9 partial derivatives ONCE per evaluation, in scalars (~190 operations
// on top of the ~150 of the value; the chain rule runs over the six quantities the tire forces actually see -- tan(alpha_f), tan(alpha_r), Fxf, Fxr, sin delta,
// cos delta -- through the three load-transfer iterations), then every direction is a 3 x 5 and three 1 x 3 products (~25 operations).  Same function, same
// derivatives; the rounding differs from the forward-mode evaluation by a few ulp.
struct TrackJac {
    real f[6];        // the right-hand side (ds', Ux', Uy', r', dpsi', e')
    real a0[3];       // d ds'  / d(Ux, Uy, dpsi)
    real a4[3];       // d dpsi'/ d(Ux, Uy, dpsi)      (d dpsi'/dr = 1)
    real a5[3];       // d e'   / d(Ux, Uy, dpsi)
    real b[3][5];     // d(Ux', Uy', r') / d(Ux, Uy, r, delta, Fx)
};
// tire force with its three partial derivatives (the forward-mode branch of fiala<T> above, as a scalar function)
PG_DEV void fiala_s(real ta, real Ca, real mu, real fx, real Fz, real& Fy, real& g_t, real& g_x, real& g_z) {
    const real fm = mu * Fz;
    if (fabs(fx) >= fm) { Fy = real(0.0); g_t = real(0.0); g_x = real(0.0); g_z = real(0.0); return; }
    const real w = fm * fm - fx * fx, z = pg_rsqrt(w);
    const real rho = fabs(ta * z) * (Ca * (real(1.0) / real(3.0)));
    real g_w;
    if (rho <= real(1.0)) {
        const real cat = Ca * ta, omr = real(1.0) - rho;
        Fy = -cat * (omr + rho * rho * (real(1.0) / real(3.0)));
        g_t = -Ca * omr * omr;
        g_w = cat * (rho * (real(2.0) / real(3.0)) - real(1.0)) * rho * (real(0.5) * z * z);
    } else {
        const real sg = sgn(ta);
        Fy = -(w * z) * sg; g_t = real(0.0); g_w = real(-0.5) * z * sg;
    }
    g_z = g_w * (real(2.0) * fm * mu); g_x = g_w * (real(-2.0) * fx);
}
// VehicleModel{TrackingBicycleModel} (vehicle_dynamics.jl:310-315 over :159-183, actuator limits :293-298, tire model :35-76) at q = (ds, Ux, Uy, r, dpsi, e),
// u = (delta, Fx), p = (V, kappa): value and Jacobian
PG_DEV void tracking_jac(const DevVehicle& P, const real q[6], real u0, real u1, real pV, real pK, TrackJac& J) {
    const real Ux = q[1], Uy = q[2], r = q[3];
    // apply_control_limits + longitudinal_tire_forces: a clamped input has no derivative (the limits see Ux by value, :295)
    const real dlt = u0 > P.delta_max ? P.delta_max : (u0 < -P.delta_max ? -P.delta_max : u0);
    const real ddl = (u0 > P.delta_max || u0 < -P.delta_max) ? real(0.0) : real(1.0);
    const real cap = jmin(P.Fx_max, P.Px_max / Ux);
    real fxc = u1, dfx = real(1.0);
    if (cap < fxc) { fxc = cap; dfx = real(0.0); }
    if (P.Fx_min > fxc) { fxc = P.Fx_min; dfx = real(0.0); }
    const real ff = fxc > real(0.0) ? P.fwd_frac : P.fwb_frac, fr = fxc > real(0.0) ? P.rwd_frac : P.rwb_frac;
    const real Xf = fxc * ff, Xr = fxc * fr;
    // kinematic rows
    real s, c; pg_sincos(q[4], &s, &c);
    const real vs = Ux * c - Uy * s, w5 = Ux * s + Uy * c;
    J.f[0] = vs - pV; J.f[4] = r - vs * pK; J.f[5] = w5;
    J.a0[0] = c; J.a0[1] = -s; J.a0[2] = -w5;
    J.a4[0] = -pK * c; J.a4[1] = pK * s; J.a4[2] = pK * w5;
    J.a5[0] = s; J.a5[1] = c; J.a5[2] = vs;
    // slip angles (tangents), drag
    real sd, cd; pg_sincos(dlt, &sd, &cd);
    const real iux = frcp(Ux), yv = Uy + P.a * r, tfv = yv * iux, tdv = sd * frcp(cd);
    const real iden = frcp(real(1.0) + tfv * tdv), id2 = iden * iden;
    const real c_y = (real(1.0) + tdv * tdv) * id2 * iux, c_u = -c_y * tfv, c_d = -(real(1.0) + tfv * tfv) * id2 * (real(1.0) + tdv * tdv);
    const real taf = (tfv - tdv) * iden;
    const real dd = -(P.Cd1 + real(2.0) * P.Cd2 * Ux), Fdrag = -P.Cd0 - Ux * (P.Cd1 + P.Cd2 * Ux);
    const real tar = (Uy - P.b * r) * iux, t_u = -tar * iux;
    // lateral forces (:64-76): three load-transfer iterations on the front axle, then the rear.  Gradients over L = (taf, Xf, Xr, sd, cd), as five scalars each.
    const real W_b = P.m * P.G * P.b, W_a = P.m * P.G * P.a, invL = real(1.0) / P.L;
    const real hL = P.h * invL, WbL = W_b * invL, WaL = W_a * invL;
    const real FxfC = Xf * cd;
    real Fx = FxfC + Xr, Fxt = FxfC, Fyf = real(0.0);
    real dX[5] = {real(0.0), cd, real(1.0), real(0.0), Xf};        // d Fx
    real dT[5] = {real(0.0), cd, real(0.0), real(0.0), Xf};        // d Fxt
    real dY[5] = {real(0.0), real(0.0), real(0.0), real(0.0), real(0.0)};      // d Fyf
#pragma unroll 1
    for (int i = 0; i < 3; i++) {
        const real Fzf = WbL - hL * Fx;
        real g_t, g_x, g_z;
        fiala_s(taf, P.Caf, P.mu, Xf, Fzf, Fyf, g_t, g_x, g_z);
        const real gz = -g_z * hL;
#pragma unroll
        for (int k = 0; k < 5; k++) dY[k] = gz * dX[k];
        dY[0] += g_t; dY[1] += g_x;
        Fxt = FxfC - Fyf * sd;
        dT[0] = -sd * dY[0]; dT[1] = cd - sd * dY[1]; dT[2] = -sd * dY[2]; dT[3] = -sd * dY[3] - Fyf; dT[4] = Xf - sd * dY[4];
        Fx = Fxt + Xr;
#pragma unroll
        for (int k = 0; k < 5; k++) dX[k] = dT[k];
        dX[2] += real(1.0);
    }
    const real Fzr = WaL + hL * Fx;
    real Fyr, h_t, h_x, h_z;
    fiala_s(tar, P.Car, P.mu, Xr, Fzr, Fyr, h_t, h_x, h_z);
    const real hz = h_z * hL;
    real dR[5];                                                     // d Fyr over L (its own slip angle: h_t)
#pragma unroll
    for (int k = 0; k < 5; k++) dR[k] = hz * dX[k];
    dR[2] += h_x;
    const real Fyf_t = Fyf * cd + Xf * sd;
    real dF[5];                                                     // d Fyf_t
#pragma unroll
    for (int k = 0; k < 5; k++) dF[k] = cd * dY[k];
    dF[1] += sd; dF[3] += Xf; dF[4] += Fyf;
    const real invm = real(1.0) / P.m, invI = real(1.0) / P.Izz;
    J.f[1] = (Fxt + Xr + Fdrag) * invm + r * Uy;
    J.f[2] = (Fyf_t + Fyr) * invm - r * Ux;
    J.f[3] = (P.a * Fyf_t - P.b * Fyr) * invI;
    // gradients of the three accelerations over L (+ the rear slip angle), then the chain to (Ux, Uy, r, delta, Fx):
    //   taf: (c_u, c_y, a c_y, c_d) ; tar: (t_u, 1/Ux, -b/Ux, 0) ; sd, cd: (0, 0, 0, cd, -sd) ; Xf, Xr: Fx through the brake / drive split
    real g1[5], g2[5], g3[5];
#pragma unroll
    for (int k = 0; k < 5; k++) { g1[k] = invm * dT[k]; g2[k] = invm * (dF[k] + dR[k]); g3[k] = invI * (P.a * dF[k] - P.b * dR[k]); }
    g1[2] += invm;
    const real r2 = invm * h_t, r3 = -invI * P.b * h_t;               // d(Uy', r') / d tar
    const real acy = P.a * c_y, biux = P.b * iux;
    J.b[0][0] = g1[0] * c_u + invm * dd;  J.b[0][1] = g1[0] * c_y + r;   J.b[0][2] = g1[0] * acy + Uy;
    J.b[1][0] = g2[0] * c_u + r2 * t_u - r; J.b[1][1] = g2[0] * c_y + r2 * iux; J.b[1][2] = g2[0] * acy - r2 * biux - Ux;
    J.b[2][0] = g3[0] * c_u + r3 * t_u;   J.b[2][1] = g3[0] * c_y + r3 * iux; J.b[2][2] = g3[0] * acy - r3 * biux;
    const real dsd = cd * ddl, dcd = -sd * ddl, dta = c_d * ddl, dxf = ff * dfx, dxr = fr * dfx;
    J.b[0][3] = g1[0] * dta + g1[3] * dsd + g1[4] * dcd;  J.b[0][4] = g1[1] * dxf + g1[2] * dxr;
    J.b[1][3] = g2[0] * dta + g2[3] * dsd + g2[4] * dcd;  J.b[1][4] = g2[1] * dxf + g2[2] * dxr;
    J.b[2][3] = g3[0] * dta + g3[3] * dsd + g3[4] * dcd;  J.b[2][4] = g3[1] * dxf + g3[2] * dxr;
}

// VehicleModel{BicycleModel}: vehicle_dynamics.jl:310-314 over :111-135; only the components the hot path reads (dUx,dUy,dr)
template <class T>
PG_DEV void world_body_rhs(const DevVehicle& P, real Ux, real Uy, real r, T u0, T u1, T& dUx, T& dUy, T& dr) {
    T d, Fxf, Fxr;
    actuate<T>(P, u0, u1, Ux, d, Fxf, Fxr);
    body_accel<T>(P, cst<T>(Ux), cst<T>(Uy), cst<T>(r), d, Fxf, Fxr, dUx, dUy, dr);
}

// VehicleModel{LateralTrackingBicycleModel}: vehicle_dynamics.jl:310-316 over :205-224.  q = (Uy, r, dpsi, e), u = (delta, Fx), p = (Ux, kappa)
// Ux is a parameter here (get_Ux = p[1], :309): the actuator limits see its VALUE only (:295) while the dynamics see the full number.
template <class T>
PG_DEV void lateral_rhs(const DevVehicle& P, const T q[4], T u0, T u1, T pUx, T pK, T out[4]) {
    T d, Fxf, Fxr;
    actuate<T>(P, u0, u1, val(pUx), d, Fxf, Fxr);
    T s, c; sincos_(q[2], s, c);
    T dUx_unused, dUy, dr;
    body_accel<T>(P, pUx, q[0], q[1], d, Fxf, Fxr, dUx_unused, dUy, dr);
    out[0] = dUy; out[1] = dr;
    out[2] = q[1] - pUx * pK;
    out[3] = pUx * s + q[0] * c;
}

// sqrt(a^2 - b^2) where b may have been CLAMPED onto +-a (a tire force on the friction circle, vehicle_dynamics.jl:356,363): the reference gets exactly 0
// there because Julia rounds every product on its own.  A fused multiply-add does not: hipcc contracts across statements, so `Frm = mu * Fzr` was fused
// into `Frm * Frm - Fxr * Fxr` with its UNROUNDED value, the difference came out as minus one rounding error, the square root as NaN, the clamps that
// use it as a bound stopped clamping, and friction-saturated braking seeds were a few percent off on a coin-flip subset of nodes (found by
// tests/test_gpu_edge_cases.py::test_friction_saturated_seeding in round 2; present since round 1).  Hence: the factored form here, and
// `fp contract(off)` in the two functions whose branches rely on exact cancellation (stable_limits, steady_state).
PG_DEV real sqrt_diff_sq(real a, real b) {
#pragma clang fp contract(off)
    return sqrt((a - b) * (a + b));
}

// stable_limits: vehicle_dynamics.jl:227-263
struct Envelope { real dmin, dmax, H[4][2], G[4]; };
PG_DEV Envelope stable_limits(const DevVehicle& B, real Ux, real Fxf, real Fxr) {
#pragma clang fp contract(off)
    real Fx = Fxf + Fxr;
    real Fzf = (B.m * B.G * B.b - B.h * Fx) / B.L, Fzr = (B.m * B.G * B.a + B.h * Fx) / B.L;
    real Ffm = B.mu * Fzf, Frm = B.mu * Fzr;
    real Fyf_max = fabs(Fxf) > Ffm ? real(0.0) : sqrt_diff_sq(Ffm, Fxf);
    real Fyr_max = fabs(Fxr) > Frm ? real(0.0) : sqrt_diff_sq(Frm, Fxr);
    real tf = real(3.0) * Fyf_max / B.Caf, tr = real(3.0) * Fyr_max / B.Car;
    real af = atan(tf), ar = atan(tr);
    Envelope o;
    real muG = B.mu * B.G, Ux2 = Ux * Ux;
    o.dmax = atan(B.L * muG / Ux2 - tr) + af;
    o.dmin = atan(-B.L * muG / Ux2 + tr) - af;
    real rC = muG / Ux, UyC = -Ux * tr + B.b * rC;
    real rD = Ux / B.L * (tan(af + o.dmax) - tr), UyD = Ux * tr + B.b * rD;
    real mCD = (rD - rC) / (UyD - UyC);
    real rE = Ux / B.L * (tan(-af + o.dmin) + tr), UyE = -Ux * tr + B.b * rE;
    real rF = -muG / Ux, UyF = Ux * tr + B.b * rF;
    real mEF = (rF - rE) / (UyF - UyE);
    o.H[0][0] = real(1.0) / Ux;  o.H[0][1] = -B.b / Ux;
    o.H[1][0] = -real(1.0) / Ux; o.H[1][1] = B.b / Ux;
    o.H[2][0] = -mCD;      o.H[2][1] = real(1.0);
    o.H[3][0] = mEF;       o.H[3][1] = -real(1.0);
    o.G[0] = ar; o.G[1] = ar; o.G[2] = rC - UyC * mCD; o.G[3] = -rF + UyF * mEF;
    return o;
}

// steady_state_estimates: vehicle_dynamics.jl:319-390.
// The fixed point of the reference re-evaluates sincos(beta) and sincos(delta) at the top of every iteration, with beta = atan(..) and
// delta = atan2(..) - atan(..) produced at the bottom of the previous one.  Here the pair (sin, cos) is carried instead of the angle:
//   beta = atan(tb)                       =>  cos beta = 1/sqrt(1 + tb^2), sin beta = tb cos beta
//   delta = atan2(y, Ux) - atan(taf)      =>  tan delta = (ty - taf)/(1 + ty taf) with ty = y/Ux (Ux = V cos beta > 0), and cos delta has the sign of
//                                             (1 + ty taf) because cos(x - y) = cos x cos y (1 + tan x tan y) with both cosines positive
// (identical values in exact arithmetic; the serial 90-iteration chain of the cold node seeding drops four transcendental calls per iteration).
// The ANGLES are formed once, at exit, with the reference's own expressions.  beta, (sb, cb), (sd, cd): the initial estimates and their sines/cosines.
// (defer = true: the three inverse tangents of the exit are left to the caller -- ang_y, ang_x, ang_t, tb carry their arguments: delta = atan2(ang_y, ang_x) - atan(ang_t),
// beta = atan(tb) when beta_is_tan.  The cold node seeding is one serial chain per instance; the angles feed nothing in it and cost a fifth of its instructions)
struct Steady { real beta, Ux, Uy, r, A, delta, Fx, ang_y, ang_x, ang_t, tb; bool beta_is_tan; };
// the friction-circle limit at the top of steady_state_estimates (vehicle_dynamics.jl:325-333): (A_tan, A_rad = V^2 kappa) pulled back onto |A| <= mu G.
// A function of its own since round 5: the cold node seeding runs (V, s) ahead on this value -- the acceleration the solve below returns unless a limit binds inside it.
PG_DEV void limit_accel(const DevVehicle& P, real V, real kappa, real& A_tan, real& A_rad) {
#pragma clang fp contract(off)
    A_rad = V * V * kappa;
    const real A_max = P.mu * P.G;
    if (hypot(A_tan, A_rad) > A_max) {
        if (fabs(A_rad) > A_max) { A_rad = A_max * sgn(A_rad); A_tan = real(0.0); }
        else A_tan = sqrt_diff_sq(A_max, A_rad) * sgn(A_tan);
    }
}
PG_DEV Steady steady_state(const DevVehicle& P, real V, real A_tan, real kappa, int num_iters, real r, real beta, real sb, real cb, real delta, real sd, real cd, real Fyf, bool defer = false) {
#pragma clang fp contract(off)
    real ang_y = real(0.0), ang_x = real(1.0), ang_t = real(0.0);
    real A_rad;
    limit_accel(P, V, kappa, A_tan, A_rad);
    real rdot = A_tan * kappa;
    real Fxr = real(0.0), Fxf = real(0.0), A_out = A_tan, tb = real(0.0);
    bool beta_is_tan = false;
    // loop-invariant reciprocals (the 90-iteration serial chain of the cold seeding pays ~100 cycles per fp64 division): the same number is used wherever the
    // quotient is used, so the exact-cancellation logic below is unaffected
    const real invL = real(1.0) / P.L, inva = real(1.0) / P.a, invm = real(1.0) / P.m, inv1ba = real(1.0) / (real(1.0) + P.b / P.a);
    const real mGa = P.m * P.G * P.a, mGb = P.m * P.G * P.b, Izz_a = P.Izz / P.a, s3Caf = real(3.0) / P.Caf, s3Car = real(3.0) / P.Car;
#pragma unroll 1
    for (int i = 1;; i++) {
        real Ux = V * cb, Uy = V * sb;
        real Fx_drag = -P.Cd0 - Ux * (P.Cd1 + P.Cd2 * Ux);
        real Ax = A_tan * cb - A_rad * sb, Ay = A_tan * sb + A_rad * cb;
        real Fx = Ax * P.m - Fx_drag;
        Fx = jmin(Fx, jmin(P.Fx_max, P.Px_max / Ux) * (P.rwd_frac + P.fwd_frac * cd) - Fyf * sd);
        real Fzr = (mGa + P.h * Fx) * invL, Fzf = (mGb - P.h * Fx) * invL;
        real Frm = P.mu * Fzr, Ffm = P.mu * Fzf;
        real frac = Fx > real(0.0) ? P.rwd_frac / (P.rwd_frac + P.fwd_frac * cd) : P.rwb_frac / (P.rwb_frac + P.fwb_frac * cd);
        Fxr = clampd((Fx + Fyf * sd) * frac, -Frm, Frm);
        real Fyr_max = sqrt_diff_sq(Frm, Fxr);
        real Fyr = clampd((Ay * P.m - rdot * Izz_a) * inv1ba, -Fyr_max, Fyr_max);
        real tanar = inv_fiala_tan3(Fyr, s3Car, Fyr_max);
        real Fxf_t = clampd(Fx - Fxr, -Ffm, Ffm);
        real Fyf_tmax = sqrt_diff_sq(Ffm, Fxf_t);
        real Fyf_t = clampd((P.b * Fyr + rdot * P.Izz) * inva, -Fyf_tmax, Fyf_tmax);
        Fxf = Fxf_t * cd + Fyf_t * sd;
        Fyf = Fyf_t * cd - Fxf_t * sd;
        real Fyf_max = sqrt_diff_sq(Ffm, Fxf);
        real taf = inv_fiala_tan3(Fyf, s3Caf, Fyf_max);                        // tan(alpha_f)
        if (i == num_iters) {
            ang_y = Uy + P.a * r; ang_x = Ux; ang_t = taf;
            if (!defer) delta = atan2(ang_y, ang_x) - atan(ang_t);                // :376-377, the angle itself
            real Ax2 = (Fxf * cd - Fyf * sd + Fxr + Fx_drag) * invm;
            real Ay2 = (Fyf * cd + Fxf * sd + Fyr) * invm;
            A_out = Ax2 * cb + Ay2 * sb;
            break;
        }
        real ty = (Uy + P.a * r) / Ux, den = real(1.0) + ty * taf, td = (ty - taf) / den;
        cd = copysign(pg_rsqrt(real(1.0) + td * td), den); sd = td * cd;       // sincos of the new delta
        tb = tanar + P.b * r / Ux; beta_is_tan = true;                         // beta = atan(tb)  (:379)
        cb = pg_rsqrt(real(1.0) + tb * tb); sb = tb * cb;
    }
    Steady o;
    o.beta = (beta_is_tan && !defer) ? atan(tb) : beta; o.ang_y = ang_y; o.ang_x = ang_x; o.ang_t = ang_t; o.tb = tb; o.beta_is_tan = beta_is_tan; o.Ux = V * cb; o.Uy = V * sb; o.r = r; o.A = A_out; o.delta = delta; o.Fx = Fxf + Fxr;
    return o;
}

// ---- trajectory (trajectories.jl) ----
struct TrajView {
    int L;
    const real *t, *s, *V, *A, *E, *N, *psi, *kappa, *edge_L, *edge_R;   // theta, phi are carried by the tube but read by nothing on this path
};
// count of elements < x  (Julia searchsortedfirst - 1)
template <class V> PG_DEV int count_less(const V* v, int n, real x) {
    int lo = 0, hi = n;
    while (lo < hi) { int mid = (lo + hi) >> 1; if (v[mid] < x) lo = mid + 1; else hi = mid; }
    return lo;
}
// count of elements <= x (Julia searchsortedlast)
template <class V> PG_DEV int count_leq(const V* v, int n, real x) {
    int lo = 0, hi = n;
    while (lo < hi) { int mid = (lo + hi) >> 1; if (v[mid] <= x) lo = mid + 1; else hi = mid; }
    return lo;
}
PG_DEV int clampi(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }
// traj(t).s : trajectories.jl:47-54 (only the arclength is read by the hot path, coupled_lat_long.jl:77,96,114)
PG_DEV real traj_s_at_time(const TrajView& T, real tq) {
    int i = clampi(count_less(T.t, T.L, tq), 1, T.L - 1) - 1;
    real Ai = (T.V[i + 1] - T.V[i]) / (T.t[i + 1] - T.t[i]);
    real dt = tq - T.t[i];
    return T.s[i] + T.V[i] * dt + Ai * dt * dt * real(0.5);
}
// traj[s] : trajectories.jl:55-68 -> (V, A, psi, kappa)
struct TrajS { real V, A, psi, kappa; };
PG_DEV TrajS traj_at_s(const TrajView& T, real sq) {
    int i = clampi(count_less(T.s, T.L, sq), 1, T.L - 1) - 1;
    real Ai = (T.V[i + 1] - T.V[i]) / (T.t[i + 1] - T.t[i]);
    real ds = sq - T.s[i], dt;
    if (fabs(Ai) < real(1e-3) || sq > T.s[T.L - 1]) dt = ds / T.V[i];
    else dt = real(2.0) * ds / (sqrt(real(2.0) * Ai * ds + T.V[i] * T.V[i]) + T.V[i]);       // (sqrt(2 A ds + V^2) - V) / A of :63, rationalised (no cancellation)
    TrajS o; o.V = T.V[i] + Ai * dt; o.A = Ai;
    int j = clampi(count_leq(T.s, T.L, sq), 1, T.L - 1) - 1;          // interp_by_s: Gridded(Linear()) + Line() (trajectories.jl:32-35)
    real w = (sq - T.s[j]) / (T.s[j + 1] - T.s[j]);
    o.psi = T.psi[j] + w * (T.psi[j + 1] - T.psi[j]);
    o.kappa = T.kappa[j] + w * (T.kappa[j + 1] - T.kappa[j]);
    return o;
}
// traj[sq] and traj(tq).s of one node in ONE loop: the two binary searches (arclength channel, time channel) advance in lockstep so that their dependent
// LDS / L2 probes overlap, and searchsortedlast on the arclength channel is derived from searchsortedfirst (the knots are strictly increasing: at most
// one equals sq) instead of being searched again.  Same results as traj_at_s + traj_s_at_time.
// (j_out, w_out: knot and weight of the interp_by_s channels at sq -- psi and kappa here, edge_L / edge_R for the caller: traj_edges_at_s at the same arclength)
PG_DEV void traj_lookup2(const TrajView& T, real sq, real tq, TrajS& o, real& s_at_t, int* j_out = nullptr, real* w_out = nullptr) {
    int lo_s = 0, hi_s = T.L, lo_t = 0, hi_t = T.L;
    while (lo_s < hi_s || lo_t < hi_t) {
        const int ms = (lo_s + hi_s) >> 1, mt = (lo_t + hi_t) >> 1;
        const real vs = T.s[ms < T.L ? ms : T.L - 1], vt = T.t[mt < T.L ? mt : T.L - 1];
        if (lo_s < hi_s) { if (vs < sq) lo_s = ms + 1; else hi_s = ms; }
        if (lo_t < hi_t) { if (vt < tq) lo_t = mt + 1; else hi_t = mt; }
    }
    {   // traj(tq).s : trajectories.jl:47-54
        const int i = clampi(lo_t, 1, T.L - 1) - 1;
        const real Ai = (T.V[i + 1] - T.V[i]) / (T.t[i + 1] - T.t[i]);
        const real dt = tq - T.t[i];
        s_at_t = T.s[i] + T.V[i] * dt + Ai * dt * dt * real(0.5);
    }
    const int i = clampi(lo_s, 1, T.L - 1) - 1;          // traj[sq] : trajectories.jl:55-68
    const real Ai = (T.V[i + 1] - T.V[i]) / (T.t[i + 1] - T.t[i]);
    const real ds = sq - T.s[i]; real dt;
    if (fabs(Ai) < real(1e-3) || sq > T.s[T.L - 1]) dt = ds / T.V[i];
    else dt = real(2.0) * ds / (sqrt(real(2.0) * Ai * ds + T.V[i] * T.V[i]) + T.V[i]);
    o.V = T.V[i] + Ai * dt; o.A = Ai;
    const int leq = lo_s + ((lo_s < T.L && T.s[lo_s < T.L ? lo_s : T.L - 1] == sq) ? 1 : 0);
    const int j = clampi(leq, 1, T.L - 1) - 1;
    const real w = (sq - T.s[j]) / (T.s[j + 1] - T.s[j]);
    o.psi = T.psi[j] + w * (T.psi[j + 1] - T.psi[j]);
    o.kappa = T.kappa[j] + w * (T.kappa[j + 1] - T.kappa[j]);
    if (j_out) { *j_out = j; *w_out = w; }
}
// edge_L, edge_R channels of interp_by_s at arclength sq (trajectories.jl:32-35): read only by the build-defined wall rows
PG_DEV void traj_edges_at_s(const TrajView& T, real sq, real& eL, real& eR) {
    int j = clampi(count_leq(T.s, T.L, sq), 1, T.L - 1) - 1;
    real w = (sq - T.s[j]) / (T.s[j + 1] - T.s[j]);
    eL = T.edge_L[j] + w * (T.edge_L[j + 1] - T.edge_L[j]); eR = T.edge_R[j] + w * (T.edge_R[j + 1] - T.edge_R[j]);
}
// adiff: DifferentialDynamicsModels (absent); semantics restated at PigeonViz.jl:24-28
PG_DEV real adiff(real x, real y) {
    const real twopi = real(6.283185307179586476925286766559);
    real d = fmod(x - y, twopi);
    if (d < real(0.0)) d += twopi;
    return d <= real(3.14159265358979323846) ? d : d - twopi;
}

}  // namespace pg
