// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the shipped product path.
// CPU restatement of /root/reference/src/trajectories.jl and src/math.jl.  Parity UNPINNED (no golden
// vectors in the reference).  The 1-D gridded-linear interpolation with Line() extrapolation restates the
// published behaviour of Interpolations.jl 0.11.2 (third-party, absent; env/Manifest.toml:133-137;
// call site trajectories.jl:32-35): i = clamp(searchsortedlast(knots,x),1,n-1), w=(x-k_i)/(k_{i+1}-k_i),
// value = c_i + w (c_{i+1}-c_i), same formula outside [k_1,k_n].
#pragma once
#include <cmath>
#include <vector>

namespace po {

struct TrajectoryNode { double t, s, V, A, E, N, psi, kappa, theta, phi, edge_L, edge_R; };

struct TrajectoryTube {
    int L = 0;
    std::vector<double> t, s, V, A, E, N, psi, kappa, theta, phi, edge_L, edge_R;   // trajectories.jl:8-21

    // Julia searchsortedfirst(v, x, 1, L, Forward): first 1-based index with v[i] >= x, L+1 if none.
    static int searchsortedfirst(const std::vector<double>& v, double x) {
        int lo = 0, hi = (int)v.size();     // 0-based half-open; returns count of elements < x
        while (lo < hi) { int mid = (lo + hi) / 2; if (v[mid] < x) lo = mid + 1; else hi = mid; }
        return lo + 1;
    }
    // searchsortedlast: last 1-based index with v[i] <= x, 0 if none.
    static int searchsortedlast(const std::vector<double>& v, double x) {
        int lo = 0, hi = (int)v.size();     // returns count of elements <= x
        while (lo < hi) { int mid = (lo + hi) / 2; if (v[mid] <= x) lo = mid + 1; else hi = mid; }
        return lo;
    }
    static int clampi(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }

    // interp_by_s (trajectories.jl:32-35): 8 spatial channels, gridded linear in s, Line() extrapolation.
    void spatial(double sq, TrajectoryNode& o) const {
        int i = clampi(searchsortedlast(s, sq), 1, L - 1) - 1;    // 0-based left knot
        double w = (sq - s[i]) / (s[i + 1] - s[i]);
        auto lerp = [&](const std::vector<double>& c) { return c[i] + w * (c[i + 1] - c[i]); };
        o.E = lerp(E); o.N = lerp(N); o.psi = lerp(psi); o.kappa = lerp(kappa);
        o.theta = lerp(theta); o.phi = lerp(phi); o.edge_L = lerp(edge_L); o.edge_R = lerp(edge_R);
    }
    // traj(t): trajectories.jl:47-54
    TrajectoryNode at_time(double tq) const {
        int i = clampi(searchsortedfirst(t, tq) - 1, 1, L - 1) - 1;
        double Ai = (V[i + 1] - V[i]) / (t[i + 1] - t[i]);
        double dt = tq - t[i];
        TrajectoryNode o;
        o.t = tq; o.s = s[i] + V[i] * dt + Ai * dt * dt / 2; o.V = V[i] + Ai * dt; o.A = Ai;
        spatial(o.s, o);
        return o;
    }
    // traj[s]: trajectories.jl:55-68
    TrajectoryNode at_s(double sq) const {
        int i = clampi(searchsortedfirst(s, sq) - 1, 1, L - 1) - 1;
        double Ai = (V[i + 1] - V[i]) / (t[i + 1] - t[i]);
        double ds = sq - s[i];
        double dt;
        if (std::fabs(Ai) < 1e-3 || sq > s[L - 1]) dt = ds / V[i];
        else dt = (std::sqrt(2 * Ai * ds + V[i] * V[i]) - V[i]) / Ai;
        TrajectoryNode o;
        o.t = t[i] + dt; o.s = sq; o.V = V[i] + Ai * dt; o.A = Ai;
        spatial(sq, o);
        return o;
    }
    // math.jl:4-9
    static double distance2(double ax, double ay, double bx, double by, double x, double y) {
        double vx = bx - ax, vy = by - ay;
        double lam = (vx * (x - ax) + vy * (y - ay)) / (vx * vx + vy * vy);
        lam = lam < 0 ? 0 : (lam > 1 ? 1 : lam);
        double px = (1 - lam) * ax + lam * bx, py = (1 - lam) * ay + lam * by;
        return (px - x) * (px - x) + (py - y) * (py - y);
    }
    // path_coordinates: trajectories.jl:71-94 (strict '<' => lowest index wins ties)
    void path_coordinates(double x, double y, double& s_out, double& e_out, double& t_out, int* imin_out = nullptr) const {
        double d2min = INFINITY; int imin = 0;
        for (int i = 0; i < L - 1; i++) {
            double d2 = distance2(E[i], N[i], E[i + 1], N[i + 1], x, y);
            if (d2 < d2min) { d2min = d2; imin = i; }
        }
        int i = imin;
        double vx = E[i + 1] - E[i], vy = N[i + 1] - N[i];
        double wx = x - E[i], wy = y - N[i];
        double ds = std::sqrt(wx * wx + wy * wy - d2min);
        s_out = s[i] + ds;
        double cr = vx * wy - vy * wx;
        e_out = std::sqrt(d2min) * ((cr > 0) - (cr < 0));
        double Ai = (V[i + 1] - V[i]) / (t[i + 1] - t[i]);
        double dt;
        if (std::fabs(Ai) < 1e-3) dt = ds / V[i];
        else dt = (std::sqrt(2 * Ai * ds + V[i] * V[i]) - V[i]) / Ai;
        t_out = t[i] + dt;
        if (imin_out) *imin_out = imin;
    }
};

// DifferentialDynamicsModels.adiff / mod2piF (third-party, absent); semantics restated in-tree at
// /root/reference/src/PigeonViz.jl:24-28:  d = mod(x - y, 2pi); d <= pi ? d : d - 2pi   (mod = floored modulo)
inline double adiff(double x, double y) {
    const double twopi = 2 * M_PI;
    double d = std::fmod(x - y, twopi);
    if (d < 0) d += twopi;
    return d <= M_PI ? d : d - twopi;
}

}  // namespace po
