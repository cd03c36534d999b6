// ORACLE (test infrastructure): Julia's floating-point RANGE arithmetic, restated from Julia 1.0 Base (base/twiceprecision.jl, base/range.jl, base/broadcast.jl) -- the reference
// builds both of its time axes out of ranges:
//     TS.ts[1:N_short+1]   .= t0      .+ dt_short*(0:N_short)        model_predictive_control.jl:25
//     TS.ts[N_short+2:end] .= t0_long .+ dt_long*(1:N_long)          model_predictive_control.jl:26
//     for t in 0:dt:mpc.trajectory.t[end]                            model_predictive_control.jl:87
// `x*(a:b)` is `range(x*a, step=x*1, length=...)`: a StepRangeLen whose reference value and step are TwicePrecision numbers, lifted to the exact rational when start and
// step have one (0.01 = 1/100, 0.2 = 1/5); `t .+ range` stays such a range (the scalar is added to the reference value in twice precision); element i is ONE rounding of
// ref + (i - offset) step.  So 0.2*(1:20)[3] == 0.6 where the naive product 0.2*3 gives 0.6000000000000001.
// PARITY UNPINNED like the rest of the oracle, and more so: Base is not part of /root/reference and no Julia is installed -- this is a reading of Julia 1.0's sources from
// memory that could not be executed.  The naive two-rounding form of rounds 1-5 stays available for A/B (MPCTimeSteps::naive_time_grid).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>

namespace jlrange {

struct Twice { double hi, lo; };

// twiceprecision.jl: canonicalize2, add12, mul12
inline Twice canonicalize2(double big, double little) { const double h = big + little; return {h, (big - h) + little}; }
inline Twice add12(double x, double y) { if (std::fabs(y) > std::fabs(x)) { const double t = x; x = y; y = t; } return canonicalize2(x, y); }
inline Twice mul12(double x, double y) {
    const double h = x * y;
    if (h == 0.0 || !std::isfinite(h)) return {h, h};
    return canonicalize2(h, std::fma(x, y, -h));
}
// truncbits: the low nb bits of the significand cleared (so that the value times an integer below 2^nb is exact)
inline double truncbits(double x, int nb) {
    if (nb <= 0) return x;
    uint64_t u; std::memcpy(&u, &x, 8);
    u &= nb >= 64 ? 0ull : (~0ull << nb);
    double r; std::memcpy(&r, &u, 8); return r;
}
// TwicePrecision{Float64}(x) / TwicePrecision{Float64}(y)
inline Twice div(Twice x, Twice y) {
    const double hi = x.hi / y.hi;
    const Twice u = mul12(hi, y.hi);
    const double lo = ((((x.hi - u.hi) - u.lo) + x.lo) - hi * y.lo) / y.hi;
    return canonicalize2(hi, lo);
}
// TwicePrecision{Float64}(i::Integer): splitprec -- the high half holds the top 27 bits (integers that matter here are far below 2^26: lo = 0)
inline Twice from_int(long long i) {
    const double hi = truncbits((double)i, 27);
    return canonicalize2(hi, (double)(i - (long long)hi));
}
inline Twice from_ratio(long long n, long long d) { return div(from_int(n), from_int(d)); }          // TwicePrecision{T}((n, d))
inline Twice twiceprecision(Twice v, int nb) { const double hi = truncbits(v.hi, nb); return {hi, (v.hi - hi) + v.lo}; }
inline Twice add(Twice x, double y) { const Twice s = add12(x.hi, y); return canonicalize2(s.hi, s.lo + x.lo); }      // +(x::TwicePrecision, y::Number)

// rat(x): continued-fraction lift with numerator and denominator bounded by maxintfloat(Float32) = 2^24 (twiceprecision.jl)
inline void rat(double x, long long& num, long long& den) {
    double y = x;
    long long a = 1, d = 1, b = 0, c = 0;
    const double m = 16777216.0;
    while (std::fabs(y) <= m) {
        const long long f = (long long)std::trunc(y);
        y -= (double)f;
        const long long a2 = f * a + c, b2 = f * b + d;
        c = a; a = a2; d = b; b = b2;
        if (!(std::llabs(a) <= (long long)m && std::llabs(b) <= (long long)m)) { num = c; den = d; return; }
        if ((double)a / (double)b == x) break;
        y = 1.0 / y;
    }
    num = a; den = b;
}
inline long long gcd_ll(long long a, long long b) { a = std::llabs(a); b = std::llabs(b); while (b) { const long long t = a % b; a = b; b = t; } return a; }
inline long long lcm_ll(long long a, long long b) { if (a == 0 || b == 0) return 0; return std::llabs(a / gcd_ll(a, b) * b); }

// nbitslen(Float64, len, offset) = min(cld(53, 2), nbitslen(len, offset)); nbitslen(len, offset) = len < 2 ? 0 : ceil(Int, log2(max(offset-1, len-offset))) + 1
inline int nbitslen(long long len, long long offset) {
    if (len < 2) return 0;
    const long long mx = offset - 1 > len - offset ? offset - 1 : len - offset;
    const int nb = (int)std::ceil(std::log2((double)mx)) + 1;
    return nb < 27 ? nb : 27;
}

// StepRangeLen{Float64, TwicePrecision, TwicePrecision}
struct Range { Twice ref, step; long long len, offset; };
inline double elem(const Range& r, long long i) {              // unsafe_getindex, i 1-based
    const double u = (double)(i - r.offset);
    const double shift_hi = u * r.step.hi, shift_lo = u * r.step.lo;
    const Twice x = add12(r.ref.hi, shift_hi);
    return x.hi + (x.lo + (shift_lo + r.ref.lo));
}
inline Range shifted(const Range& r, double x) { Range o = r; o.ref = add(r.ref, x); return o; }      // broadcasted(+, x::Number, r::StepRangeLen)

inline Range steprangelen_hp_ratio(long long ref_n, long long step_n, long long den, int nb, long long len, long long offset) {
    return Range{from_ratio(ref_n, den), twiceprecision(from_ratio(step_n, den), nb), len, offset};
}
inline Range steprangelen_hp_literal(double ref, double step, int nb, long long len, long long offset) {
    return Range{Twice{ref, 0.0}, twiceprecision(Twice{step, 0.0}, nb), len, offset};
}
inline Range floatrange(long long start_n, long long step_n, long long len, long long den) {
    if (len < 2 || step_n == 0) return steprangelen_hp_ratio(start_n, step_n, den, 0, len, 1);
    long long imin = (long long)std::nearbyint(-(double)start_n / (double)step_n + 1.0);      // round(Int, x): ties to even, like nearbyint in the default rounding mode
    imin = imin < 1 ? 1 : (imin > len ? len : imin);
    const long long ref_n = start_n + (imin - 1) * step_n;
    return steprangelen_hp_ratio(ref_n, step_n, den, nbitslen(len, imin), len, imin);
}
const double MAXINTFLOAT64 = 9007199254740992.0;
// range(a, step = st, length = len) for Float64 (_range(a, st, nothing, len))
inline Range range_start_step_len(double a, double st, long long len) {
    long long sn, sd, tn, td;
    rat(a, sn, sd); rat(st, tn, td);
    if (sd != 0 && td != 0 && (double)sn / (double)sd == a && (double)tn / (double)td == st) {
        const long long den = lcm_ll(sd, td);
        if (std::fabs((double)den * a) <= MAXINTFLOAT64 && std::fabs((double)den * st) <= MAXINTFLOAT64 && den % sd == 0 && den % td == 0)
            return floatrange((long long)std::nearbyint((double)den * a), (long long)std::nearbyint((double)den * st), len, den);
    }
    return steprangelen_hp_literal(a, st, 0, len, 1);
}
inline Range scalar_times_unitrange(double x, long long first, long long last) {       // x*(first:last)
    const long long len = last >= first ? last - first + 1 : 0;
    return range_start_step_len(x * (double)first, x * 1.0, len);
}
inline bool isbetween(double a, double x, double b) { return (a <= x && x <= b) || (b <= x && x <= a); }
// start:step:stop for Float64 (range.jl / twiceprecision.jl `(:)`)
inline Range colon(double start, double step, double stop) {
    long long tn, td;
    rat(step, tn, td);
    if (td != 0 && (double)tn / (double)td == step) {
        long long sn, sd, en, ed;
        rat(start, sn, sd); rat(stop, en, ed);
        if (sd != 0 && ed != 0 && (double)sn / (double)sd == start && (double)en / (double)ed == stop) {
            const long long den = lcm_ll(sd, td);
            if (den != 0 && std::fabs(start * (double)den) <= MAXINTFLOAT64 && std::fabs(step * (double)den) <= MAXINTFLOAT64 && den % sd == 0 && den % td == 0) {
                const long long start_n = (long long)std::nearbyint(start * (double)den), step_n = (long long)std::nearbyint(step * (double)den);
                long long len = (den * en - ed * start_n + step_n * ed) / (step_n * ed);      // div: truncation, as Julia's
                if (len < 0) len = 0;
                if (isbetween(start, start + (double)(len - 1) * step, stop + step / 2) && !isbetween(start, start + (double)len * step, stop))
                    return floatrange(start_n, step_n, len, den);
            }
        }
    }
    const double lf = (stop - start) / step;
    long long len;
    if (lf < 0) len = 0;
    else if (lf == 0) len = 1;
    else {
        len = (long long)std::nearbyint(lf) + 1;
        const double stop2 = start + (double)(len - 1) * step;
        len -= ((start < stop && stop < stop2) ? 1 : 0) + ((start > stop && stop > stop2) ? 1 : 0);
    }
    return steprangelen_hp_literal(start, step, 0, len, 1);
}

}  // namespace jlrange
