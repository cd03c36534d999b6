// Julia's floating-point range arithmetic for the two time axes of the reference (round 6).
//   model_predictive_control.jl:25-26   ts[1:Ns+1] .= t0 .+ dt_short*(0:Ns);   ts[Ns+2:end] .= t0_long .+ dt_long*(1:Nl)
//   model_predictive_control.jl:87      for t in 0:dt:mpc.trajectory.t[end]
// In Julia `x*(a:b)` and `a:s:b` are StepRangeLen ranges whose reference value and step are kept in twice the working precision (Base twiceprecision.jl), lifted to the exact
// rational when start and step have one (0.01 = 1/100, 0.2 = 1/5), and `t .+ range` is again such a range: element i is ONE rounding of ref + (i - offset) step, where
// `t0 + dt*i` rounds twice.  The difference is one ulp of ts -- but `ceil((t0_long + dt_short)/dt_long - 1)` (:23) is discontinuous and `simulate` from t = 0 with dt = 0.01
// lands on that lattice every twentieth step.  The range parameters depend on the configuration only: built on the HOST (pg_create / pg_simulate_dev), evaluated on the device.
// This restates Julia 1.0's Base from memory -- Base is not under the reference tree and no Julia is installed: a reading that could not be executed.
// Option "time_grid_naive" = 1 selects the two-rounding form of rounds 1-5.
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

namespace pg {

struct JlRange { double ref_hi, ref_lo, step_hi, step_lo; int offset, len; };      // StepRangeLen{Float64, TwicePrecision, TwicePrecision}

#ifndef PG_HD
#define PG_HD __host__ __device__ inline
#endif

// error-free sum of two doubles (add12 / canonicalize2 of twiceprecision.jl)
PG_HD void jl_add12(double x, double y, double& h, double& l) {
#pragma clang fp contract(off)
    const bool sw = fabs(y) > fabs(x);
    const double a = sw ? y : x, b = sw ? x : y;
    h = a + b; l = (a - h) + b;
}
// (t .+ r)[i], i 1-based: the scalar joins the reference value in twice precision (broadcasted(+, x, r::StepRangeLen)), then unsafe_getindex
PG_HD double jl_shifted_elem(const JlRange& r, double t, int i) {
#pragma clang fp contract(off)
    double s_hi, s_lo, rh, rl;
    jl_add12(r.ref_hi, t, s_hi, s_lo);
    { const double little = s_lo + r.ref_lo; rh = s_hi + little; rl = (s_hi - rh) + little; }      // canonicalize2(s_hi, s_lo + ref.lo)
    const double u = (double)(i - r.offset);
    const double shift_hi = u * r.step_hi, shift_lo = u * r.step_lo;
    double x_hi, x_lo;
    jl_add12(rh, shift_hi, x_hi, x_lo);
    return x_hi + (x_lo + (shift_lo + rl));
}

// ---- construction (host) ----
namespace jl_host {
struct TP { double hi, lo; };
inline TP canon2(double big, double little) { const double h = big + little; return TP{h, (big - h) + little}; }
inline TP mul12(double x, double y) { const double h = x * y; if (h == 0.0 || !isfinite(h)) return TP{h, h}; return canon2(h, fma(x, y, -h)); }
inline double truncbits(double x, int nb) { if (nb <= 0) return x; uint64_t u; memcpy(&u, &x, 8); u &= nb >= 64 ? 0ull : (~0ull << nb); double r; memcpy(&r, &u, 8); return r; }
inline TP of_int(long long i) { const double hi = truncbits((double)i, 27); return canon2(hi, (double)(i - (long long)hi)); }
inline TP quot(TP x, TP y) {
#pragma clang fp contract(off)
    const double hi = x.hi / y.hi; const TP u = mul12(hi, y.hi); return canon2(hi, ((((x.hi - u.hi) - u.lo) + x.lo) - hi * y.lo) / y.hi); }
inline TP trunc_hi(TP v, int nb) { const double hi = truncbits(v.hi, nb); return TP{hi, (v.hi - hi) + v.lo}; }
inline void rat(double x, long long& n, long long& d_) {      // continued fraction, terms bounded by maxintfloat(Float32)
    double y = x; long long a = 1, d = 1, b = 0, c = 0; const long long m = 1 << 24;
    while (fabs(y) <= (double)m) {
        const long long f = (long long)trunc(y); y -= (double)f;
        const long long an = f * a + c, bn = f * b + d; c = a; a = an; d = b; b = bn;
        if (!(llabs(a) <= m && llabs(b) <= m)) { n = c; d_ = d; return; }
        if ((double)a / (double)b == x) break;
        y = 1.0 / y;
    }
    n = a; d_ = b;
}
inline long long gcd(long long a, long long b) { a = llabs(a); b = llabs(b); while (b) { const long long t = a % b; a = b; b = t; } return a; }
inline int nbits(long long len, long long off) { if (len < 2) return 0; const long long mx = off - 1 > len - off ? off - 1 : len - off; const int nb = (int)ceil(log2((double)mx)) + 1; return nb < 27 ? nb : 27; }
inline JlRange make(TP ref, TP step, long long len, long long off) { return JlRange{ref.hi, ref.lo, step.hi, step.lo, (int)off, (int)len}; }
inline JlRange literal(double a, double st, long long len) { return make(TP{a, 0.0}, TP{st, 0.0}, len, 1); }
inline JlRange floatrange(long long start_n, long long step_n, long long len, long long den) {
    if (len < 2 || step_n == 0) return make(quot(of_int(start_n), of_int(den)), quot(of_int(step_n), of_int(den)), len, 1);
    long long imin = (long long)nearbyint(-(double)start_n / (double)step_n + 1.0);
    imin = imin < 1 ? 1 : (imin > len ? len : imin);
    return make(quot(of_int(start_n + (imin - 1) * step_n), of_int(den)), trunc_hi(quot(of_int(step_n), of_int(den)), nbits(len, imin)), len, imin);
}
const double MAXINT = 9007199254740992.0;
}  // namespace jl_host

// x*(first:last) = range(x*first, step = x, length = last - first + 1)
inline JlRange jl_scalar_times_unitrange(double x, long long first, long long last) {
#pragma clang fp contract(off)      // (hipcc contracts host code too: every rounding here is Julia's)
    using namespace jl_host;
    const long long len = last >= first ? last - first + 1 : 0;
    const double a = x * (double)first, st = x * 1.0;
    long long sn, sd, tn, td; rat(a, sn, sd); rat(st, tn, td);
    if (sd != 0 && td != 0 && (double)sn / (double)sd == a && (double)tn / (double)td == st) {
        const long long den = llabs(sd / gcd(sd, td) * td);
        if (fabs((double)den * a) <= MAXINT && fabs((double)den * st) <= MAXINT && den % sd == 0 && den % td == 0)
            return floatrange((long long)nearbyint((double)den * a), (long long)nearbyint((double)den * st), len, den);
    }
    return literal(a, st, len);
}
// start:step:stop
inline JlRange jl_colon(double start, double step, double stop) {
#pragma clang fp contract(off)
    using namespace jl_host;
    auto between = [](double a, double x, double b) { return (a <= x && x <= b) || (b <= x && x <= a); };
    long long tn, td; rat(step, tn, td);
    if (td != 0 && (double)tn / (double)td == step) {
        long long sn, sd, en, ed; rat(start, sn, sd); rat(stop, en, ed);
        if (sd != 0 && ed != 0 && (double)sn / (double)sd == start && (double)en / (double)ed == stop) {
            const long long den = llabs(sd / gcd(sd, td) * td);
            if (den != 0 && fabs(start * (double)den) <= MAXINT && fabs(step * (double)den) <= MAXINT && den % sd == 0 && den % td == 0) {
                const long long start_n = (long long)nearbyint(start * (double)den), step_n = (long long)nearbyint(step * (double)den);
                long long len = (den * en - ed * start_n + step_n * ed) / (step_n * ed);
                if (len < 0) len = 0;
                if (between(start, start + (double)(len - 1) * step, stop + step / 2) && !between(start, start + (double)len * step, stop)) return floatrange(start_n, step_n, len, den);
            }
        }
    }
    const double lf = (stop - start) / step;
    long long len;
    if (lf < 0) len = 0; else if (lf == 0) len = 1;
    else { len = (long long)nearbyint(lf) + 1; const double s2 = start + (double)(len - 1) * step; len -= ((start < stop && stop < s2) ? 1 : 0) + ((start > stop && stop > s2) ? 1 : 0); }
    return literal(start, step, len);
}

}  // namespace pg
