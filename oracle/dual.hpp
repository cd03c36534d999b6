// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the shipped product path.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under oracle/.
//
// Forward-mode dual numbers standing in for ForwardDiff.Dual (ForwardDiff 0.10.3, third-party, absent
// from /root/reference; call sites: src/vehicle_dynamics.jl:295, src/HJI_computation.jl:167 and inside
// LinearDynamicsModels.linearize, call sites src/coupled_lat_long.jl:253,262,336,348).
// Comparisons act on the value only, exactly like ForwardDiff (branches are taken on primal values).
#pragma once
#include <cmath>

namespace po {

template <int N>
struct Dual {
    double v;
    double d[N];
    Dual() : v(0) { for (int i = 0; i < N; i++) d[i] = 0; }
    Dual(double x) : v(x) { for (int i = 0; i < N; i++) d[i] = 0; }
    static Dual seed(double x, int k) { Dual r(x); r.d[k] = 1.0; return r; }
};

template <int N> inline Dual<N> operator+(const Dual<N>& a, const Dual<N>& b) { Dual<N> r; r.v = a.v + b.v; for (int i = 0; i < N; i++) r.d[i] = a.d[i] + b.d[i]; return r; }
template <int N> inline Dual<N> operator-(const Dual<N>& a, const Dual<N>& b) { Dual<N> r; r.v = a.v - b.v; for (int i = 0; i < N; i++) r.d[i] = a.d[i] - b.d[i]; return r; }
template <int N> inline Dual<N> operator-(const Dual<N>& a) { Dual<N> r; r.v = -a.v; for (int i = 0; i < N; i++) r.d[i] = -a.d[i]; return r; }
template <int N> inline Dual<N> operator*(const Dual<N>& a, const Dual<N>& b) { Dual<N> r; r.v = a.v * b.v; for (int i = 0; i < N; i++) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
template <int N> inline Dual<N> operator/(const Dual<N>& a, const Dual<N>& b) {
    Dual<N> r; double inv = 1.0 / b.v; r.v = a.v * inv;
    for (int i = 0; i < N; i++) r.d[i] = (a.d[i] - r.v * b.d[i]) * inv;
    return r;
}
template <int N> inline Dual<N> operator+(const Dual<N>& a, double b) { Dual<N> r = a; r.v += b; return r; }
template <int N> inline Dual<N> operator+(double a, const Dual<N>& b) { return b + a; }
template <int N> inline Dual<N> operator-(const Dual<N>& a, double b) { Dual<N> r = a; r.v -= b; return r; }
template <int N> inline Dual<N> operator-(double a, const Dual<N>& b) { return (-b) + a; }
template <int N> inline Dual<N> operator*(const Dual<N>& a, double b) { Dual<N> r; r.v = a.v * b; for (int i = 0; i < N; i++) r.d[i] = a.d[i] * b; return r; }
template <int N> inline Dual<N> operator*(double a, const Dual<N>& b) { return b * a; }
template <int N> inline Dual<N> operator/(const Dual<N>& a, double b) { return a * (1.0 / b); }
template <int N> inline Dual<N> operator/(double a, const Dual<N>& b) { return Dual<N>(a) / b; }

template <int N> inline bool operator<(const Dual<N>& a, const Dual<N>& b) { return a.v < b.v; }
template <int N> inline bool operator<(const Dual<N>& a, double b) { return a.v < b; }
template <int N> inline bool operator>(const Dual<N>& a, const Dual<N>& b) { return a.v > b.v; }
template <int N> inline bool operator>(const Dual<N>& a, double b) { return a.v > b; }
template <int N> inline bool operator<=(const Dual<N>& a, const Dual<N>& b) { return a.v <= b.v; }
template <int N> inline bool operator<=(const Dual<N>& a, double b) { return a.v <= b; }
template <int N> inline bool operator>=(const Dual<N>& a, const Dual<N>& b) { return a.v >= b.v; }
template <int N> inline bool operator>=(const Dual<N>& a, double b) { return a.v >= b; }

inline double value(double x) { return x; }
template <int N> inline double value(const Dual<N>& x) { return x.v; }

template <int N> inline Dual<N> chain(const Dual<N>& a, double f, double df) { Dual<N> r; r.v = f; for (int i = 0; i < N; i++) r.d[i] = df * a.d[i]; return r; }

using std::sin; using std::cos; using std::tan; using std::atan; using std::atan2; using std::sqrt; using std::fabs;
template <int N> inline Dual<N> sin(const Dual<N>& a) { return chain(a, std::sin(a.v), std::cos(a.v)); }
template <int N> inline Dual<N> cos(const Dual<N>& a) { return chain(a, std::cos(a.v), -std::sin(a.v)); }
template <int N> inline Dual<N> tan(const Dual<N>& a) { double t = std::tan(a.v); return chain(a, t, 1.0 + t * t); }
template <int N> inline Dual<N> atan(const Dual<N>& a) { return chain(a, std::atan(a.v), 1.0 / (1.0 + a.v * a.v)); }
template <int N> inline Dual<N> sqrt(const Dual<N>& a) { double s = std::sqrt(a.v); return chain(a, s, 0.5 / s); }
template <int N> inline Dual<N> atan2(const Dual<N>& y, const Dual<N>& x) {
    Dual<N> r; double den = x.v * x.v + y.v * y.v; r.v = std::atan2(y.v, x.v);
    for (int i = 0; i < N; i++) r.d[i] = (x.v * y.d[i] - y.v * x.d[i]) / den;
    return r;
}
inline double absv(double a) { return std::fabs(a); }
template <int N> inline Dual<N> absv(const Dual<N>& a) { return a.v < 0 ? -a : a; }   // ForwardDiff: d|x| = sign(x) dx
inline double signv(double a) { return (a > 0) - (a < 0); }
template <int N> inline double signv(const Dual<N>& a) { return (a.v > 0) - (a.v < 0); }   // derivative of sign is 0

}  // namespace po
