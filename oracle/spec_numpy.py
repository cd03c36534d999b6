"""INDEPENDENT numpy specification of the coupled MPC hot path (test infrastructure; never imported by the product).

Written from the Julia sources of the reference, NOT from the C++ restatement in this directory (oracle/*.hpp) and not from the HIP kernels: a second
reading of the same files by a different route (scalar Python + a small forward-mode class), so that a misreading shared by the C++ oracle and the
kernels (which one hand wrote) does not go unnoticed.  tests/test_spec_numpy.py requires the two restatements to agree to 1e-10 on the golden cases.
Parity is still unpinned against Julia itself (no Julia here, no golden vectors upstream: oracle/qp.hpp header).

Each function cites the reference lines it follows (relative to /root/reference/src).  Third-party pieces absent from the repository are restated
from their published behaviour and marked (3P): Interpolations.jl gridded-linear + Line() extrapolation, LinearDynamicsModels.linearize (RK4, 10
sub-steps, forward-mode derivatives through the integrator), DifferentialDynamicsModels.adiff.
"""
import math

import numpy as np


# ---------------------------------------------------------------------------------------------------------------------
# forward-mode number: value + gradient vector (stands in for ForwardDiff.Dual)
class Dual:
    __slots__ = ("v", "g")

    def __init__(self, v, g):
        self.v = float(v); self.g = np.asarray(g, dtype=np.float64)

    @staticmethod
    def lift(x, n):
        return x if isinstance(x, Dual) else Dual(x, np.zeros(n))

    def _o(self, o):
        return o if isinstance(o, Dual) else Dual(o, np.zeros_like(self.g))

    def __add__(self, o): o = self._o(o); return Dual(self.v + o.v, self.g + o.g)
    __radd__ = __add__
    def __sub__(self, o): o = self._o(o); return Dual(self.v - o.v, self.g - o.g)
    def __rsub__(self, o): o = self._o(o); return Dual(o.v - self.v, o.g - self.g)
    def __neg__(self): return Dual(-self.v, -self.g)
    def __mul__(self, o): o = self._o(o); return Dual(self.v * o.v, self.g * o.v + self.v * o.g)
    __rmul__ = __mul__
    def __truediv__(self, o): o = self._o(o); q = self.v / o.v; return Dual(q, (self.g - q * o.g) / o.v)
    def __rtruediv__(self, o): return self._o(o) / self
    def __abs__(self): return -self if self.v < 0 else self
    def __lt__(self, o): return self.v < val(o)
    def __le__(self, o): return self.v <= val(o)
    def __gt__(self, o): return self.v > val(o)
    def __ge__(self, o): return self.v >= val(o)


def val(x):
    return x.v if isinstance(x, Dual) else float(x)


def _lift1(f, df):
    def h(x):
        if isinstance(x, Dual):
            return Dual(f(x.v), df(x.v) * x.g)
        return f(x)
    return h


sin = _lift1(math.sin, math.cos)
cos = _lift1(math.cos, lambda v: -math.sin(v))
tan = _lift1(math.tan, lambda v: 1.0 + math.tan(v) ** 2)
sqrt = _lift1(math.sqrt, lambda v: 0.5 / math.sqrt(v))
atan = _lift1(math.atan, lambda v: 1.0 / (1.0 + v * v))


def atan2(y, x):
    if isinstance(y, Dual) or isinstance(x, Dual):
        n = len(y.g) if isinstance(y, Dual) else len(x.g)
        y = Dual.lift(y, n); x = Dual.lift(x, n)
        d = x.v * x.v + y.v * y.v
        return Dual(math.atan2(y.v, x.v), (x.v * y.g - y.v * x.g) / d)
    return math.atan2(y, x)


def sign(x):
    v = val(x)
    return (v > 0) - (v < 0)


def jl_min(a, b):      # Julia min/max propagate NaN
    return float("nan") if (a != a or b != b) else min(a, b)


def jl_max(a, b):
    return float("nan") if (a != a or b != b) else max(a, b)


# ---------------------------------------------------------------------------------------------------------------------
def X1():
    """vehicles.jl:1-59 (the entries the path reads)"""
    d = {}
    d["G"] = 9.80665
    mfl, mfr, mrl, mrr = 484, 455, 521, 504
    d["m"] = mfl + mfr + mrl + mrr
    d["Izz"] = 2900
    d["L"] = 2.87
    d["a"] = (mrl + mrr) / d["m"] * d["L"]
    d["b"] = (mfl + mfr) / d["m"] * d["L"]
    d["h"] = 0.1 * d["b"] / d["L"] + 0.1 * d["a"] / d["L"] + 0.37
    d["mu"] = 0.92; d["Caf"] = 150e3; d["Car"] = 220e3
    d["Fx_max"] = 5600; d["Px_max"] = 75e3
    d["Cd0"] = 241.0; d["Cd1"] = 25.1; d["Cd2"] = 0.0
    d["fwd_frac"] = 0.0; d["rwd_frac"] = 1 - d["fwd_frac"]; d["fwb_frac"] = 0.6; d["rwb_frac"] = 1 - d["fwb_frac"]
    d["Fx_min"] = max(-d["m"] * d["G"] * d["a"] * d["mu"] / (d["L"] * d["rwb_frac"] + d["mu"] * d["h"]),
                      -d["m"] * d["G"] * d["b"] * d["mu"] / (d["L"] * d["fwb_frac"] - d["mu"] * d["h"]))
    d["delta_max"] = 18 * math.pi / 180
    d["kappa_max"] = math.tan(d["delta_max"]) / d["L"]
    return d


def coupled_control_params():
    """coupled_lat_long.jl:23-40"""
    return dict(V_min=1.0, V_max=15.0, k_V=10 / 4 / 100, k_s=10 / 4 / 10000, deltadot_max=0.344, Q_ds=1.0, Q_dpsi=1.0, Q_e=1.0,
                W_beta=50 / (10 * math.pi / 180), W_r=50.0, W_HJI=500.0, N_HJI=3, R_delta=0.0, R_ddelta=0.1, R_Fx=0.0, R_dFx=0.5)


# ---------------------------------------------------------------------------------------------------------------------
# tire model: vehicle_dynamics.jl:35-62
def _fiala(tana, Ca, Fy_max):
    slide = 3 * Fy_max / Ca
    ratio = abs(tana / slide)
    if val(ratio) <= 1:
        return -Ca * tana * (1 - ratio + ratio * ratio / 3)
    return -Fy_max * sign(tana)


def fiala(alpha, Ca, mu, Fx, Fz):
    F_max = mu * Fz
    if abs(val(Fx)) >= val(F_max):
        return F_max * 0.0
    return _fiala(tan(alpha), Ca, sqrt(F_max * F_max - Fx * Fx))


def _inv_fiala(Fy, Ca, Fy_max):
    if abs(Fy) >= Fy_max:
        return -(3 * Fy_max / Ca) * sign(Fy)
    return -(1 + np.cbrt(abs(Fy) / Fy_max - 1)) * sign(Fy)


def lateral_tire_forces(P, af, ar, Fxf, Fxr, sd, cd, num_iters=3):
    """vehicle_dynamics.jl:64-76"""
    Fyf = Fxf * 0.0
    Fx = Fxf * cd - Fyf * sd + Fxr
    for _ in range(num_iters):
        Fzf = (P["m"] * P["G"] * P["b"] - P["h"] * Fx) / P["L"]
        Fyf = fiala(af, P["Caf"], P["mu"], Fxf, Fzf)
        Fx = Fxf * cd - Fyf * sd + Fxr
    Fzr = (P["m"] * P["G"] * P["a"] + P["h"] * Fx) / P["L"]
    Fyr = fiala(ar, P["Car"], P["mu"], Fxr, Fzr)
    return Fyf, Fyr


def longitudinal_tire_forces(P, Fx):
    """vehicle_dynamics.jl:279-283"""
    if val(Fx) > 0:
        return Fx * P["fwd_frac"], Fx * P["rwd_frac"]
    return Fx * P["fwb_frac"], Fx * P["rwb_frac"]


def apply_control_limits(P, delta, Fx, Ux):
    """vehicle_dynamics.jl:293-298 (Ux enters by VALUE; clamp/min/max pick one argument, derivative and all)"""
    Uxv = val(Ux)
    d = delta
    if val(d) < -P["delta_max"]: d = -P["delta_max"]
    if val(d) > P["delta_max"]: d = P["delta_max"]
    f = Fx
    if P["Fx_max"] < val(f): f = P["Fx_max"]
    if P["Px_max"] / Uxv < val(f): f = P["Px_max"] / Uxv
    if P["Fx_min"] > val(f): f = P["Fx_min"]
    return d, f


def _body(P, Ux, Uy, r, delta, Fxf, Fxr):
    sd, cd = sin(delta), cos(delta)
    af = atan2(Uy + P["a"] * r, Ux) - delta
    ar = atan2(Uy - P["b"] * r, Ux)
    Fyf, Fyr = lateral_tire_forces(P, af, ar, Fxf, Fxr, sd, cd)
    Fx_drag = -P["Cd0"] - Ux * (P["Cd1"] + P["Cd2"] * Ux)
    Fxf_t = Fxf * cd - Fyf * sd
    Fyf_t = Fyf * cd + Fxf * sd
    return ((Fxf_t + Fxr + Fx_drag) / P["m"] + r * Uy, (Fyf_t + Fyr) / P["m"] - r * Ux, (P["a"] * Fyf_t - P["b"] * Fyr) / P["Izz"])


def tracking_vehicle_model(P, q, u2, p):
    """VehicleModel{TrackingBicycleModel}: vehicle_dynamics.jl:310-315 over :159-183.  q = (ds, Ux, Uy, r, dpsi, e), u2 = (delta, Fx), p = (V, kappa, ., .)"""
    d, f = apply_control_limits(P, u2[0], u2[1], q[1])
    Fxf, Fxr = longitudinal_tire_forces(P, f)
    s, c = sin(q[4]), cos(q[4])
    dUx, dUy, dr = _body(P, q[1], q[2], q[3], d, Fxf, Fxr)
    vs = q[1] * c - q[2] * s
    return [vs - p[0], dUx, dUy, dr, q[3] - vs * p[1], q[1] * s + q[2] * c]


def world_vehicle_model(P, q, u2):
    """VehicleModel{BicycleModel}: vehicle_dynamics.jl:310-314 over :111-135.  q = (E, N, psi, Ux, Uy, r)"""
    d, f = apply_control_limits(P, u2[0], u2[1], q[3])
    Fxf, Fxr = longitudinal_tire_forces(P, f)
    s, c = sin(q[2]), cos(q[2])
    dUx, dUy, dr = _body(P, q[3], q[4], q[5], d, Fxf, Fxr)
    return [-q[3] * s - q[4] * c, q[3] * c - q[4] * s, q[5], dUx, dUy, dr]


def stable_limits(P, Ux, Fxf, Fxr):
    """vehicle_dynamics.jl:227-263"""
    L, b, h, m, mu, G = P["L"], P["b"], P["h"], P["m"], P["mu"], P["G"]
    Fx = Fxf + Fxr
    Fzf = (m * G * b - h * Fx) / L
    Fzr = (m * G * P["a"] + h * Fx) / L
    Ff, Fr = mu * Fzf, mu * Fzr
    Fyf_max = 0.0 if abs(Fxf) > Ff else math.sqrt(Ff * Ff - Fxf * Fxf)
    Fyr_max = 0.0 if abs(Fxr) > Fr else math.sqrt(Fr * Fr - Fxr * Fxr)
    tf, tr = 3 * Fyf_max / P["Caf"], 3 * Fyr_max / P["Car"]
    af, ar = math.atan(tf), math.atan(tr)
    d_max = math.atan(L * mu * G / (Ux * Ux) - tr) + af
    d_min = math.atan(-L * mu * G / (Ux * Ux) + tr) - af
    rC = mu * G / Ux; UyC = -Ux * tr + b * rC
    rD = Ux / L * (math.tan(af + d_max) - tr); UyD = Ux * tr + b * rD
    mCD = (rD - rC) / (UyD - UyC)
    rE = Ux / L * (math.tan(-af + d_min) + tr); UyE = -Ux * tr + b * rE
    rF = -mu * G / Ux; UyF = Ux * tr + b * rF
    mEF = (rF - rE) / (UyF - UyE)
    H = np.array([[1 / Ux, -b / Ux], [-1 / Ux, b / Ux], [-mCD, 1.0], [mEF, -1.0]])
    Gv = np.array([ar, ar, rC - UyC * mCD, -rF + UyF * mEF])
    return d_min, d_max, H, Gv


SATURATION_MARGINS = None      # set to a list to record, per steady_state_estimates call, how close the front force came to the jump of _inv_fiala


def steady_state_estimates(P, V, A_tan, kappa, num_iters=4, r=None, beta0=0.0, delta0=0.0, Fyf0=0.0):
    """vehicle_dynamics.jl:319-390"""
    r = V * kappa if r is None else r
    L, a, b, h, m, Izz, mu, G = P["L"], P["a"], P["b"], P["h"], P["m"], P["Izz"], P["mu"], P["G"]
    A_rad = V * V * kappa
    A_max = mu * G
    if math.hypot(A_tan, A_rad) > A_max:
        if abs(A_rad) > A_max:
            A_rad = A_max * sign(A_rad); A_tan = 0.0
        else:
            A_tan = math.sqrt(A_max * A_max - A_rad * A_rad) * sign(A_tan)
    rdot = A_tan * kappa
    i = 1
    beta, delta, Fyf = beta0, delta0, Fyf0
    clamp = lambda x, lo, hi: max(lo, min(hi, x))
    while True:
        sb, cb = math.sin(beta), math.cos(beta)
        sd, cd = math.sin(delta), math.cos(delta)
        Ux, Uy = V * cb, V * sb
        Fx_drag = -P["Cd0"] - Ux * (P["Cd1"] + P["Cd2"] * Ux)
        Ax = A_tan * cb - A_rad * sb
        Ay = A_tan * sb + A_rad * cb
        Fx = Ax * m - Fx_drag
        Fx = jl_min(Fx, jl_min(P["Fx_max"], P["Px_max"] / Ux) * (P["rwd_frac"] + P["fwd_frac"] * cd) - Fyf * sd)
        Fzr, Fzf = (m * G * a + h * Fx) / L, (m * G * b - h * Fx) / L
        Fr, Ff = mu * Fzr, mu * Fzf
        frac = P["rwd_frac"] / (P["rwd_frac"] + P["fwd_frac"] * cd) if Fx > 0 else P["rwb_frac"] / (P["rwb_frac"] + P["fwb_frac"] * cd)
        Fxr = clamp((Fx + Fyf * sd) * frac, -Fr, Fr)
        Fyr_max = math.sqrt(Fr * Fr - Fxr * Fxr)
        Fyr = clamp((Ay * m - rdot * Izz / a) / (1 + b / a), -Fyr_max, Fyr_max)
        tan_ar = _inv_fiala(Fyr, P["Car"], Fyr_max)
        Fxf_t = clamp(Fx - Fxr, -Ff, Ff)
        Fyf_tmax = math.sqrt(Ff * Ff - Fxf_t * Fxf_t)
        Fyf_t = clamp((b * Fyr + rdot * Izz) / a, -Fyf_tmax, Fyf_tmax)
        Fxf = Fxf_t * cd + Fyf_t * sd
        Fyf = Fyf_t * cd - Fxf_t * sd
        Fyf_max = math.sqrt(Ff * Ff - Fxf * Fxf)
        if SATURATION_MARGINS is not None:
            SATURATION_MARGINS.append(abs(abs(Fyf) - Fyf_max) / Ff)
        af = math.atan(_inv_fiala(Fyf, P["Caf"], Fyf_max))
        delta = math.atan2(Uy + a * r, Ux) - af
        if i == num_iters:
            Ax = (Fxf * cd - Fyf * sd + Fxr + Fx_drag) / m
            Ay = (Fyf * cd + Fxf * sd + Fyr) / m
            A_tan = Ax * cb + Ay * sb
            break
        i += 1
        beta = math.atan(tan_ar + b * r / Ux)
    sb, cb = math.sin(beta), math.cos(beta)
    return dict(beta=beta, Ux=V * cb, Uy=V * sb, r=r, A=A_tan, delta=delta, Fxf=Fxf, Fxr=Fxr)


# ---------------------------------------------------------------------------------------------------------------------
def adiff(x, y):
    """(3P) DifferentialDynamicsModels.adiff; semantics restated in-tree at PigeonViz.jl:24-28"""
    d = math.fmod(x - y, 2 * math.pi)
    if d < 0:
        d += 2 * math.pi
    return d if d <= math.pi else d - 2 * math.pi


class Trajectory:
    """trajectories.jl:8-94 over the 12 channel arrays (t, s, V, A, E, N, psi, kappa, theta, phi, edge_L, edge_R)"""

    def __init__(self, data12):
        (self.t, self.s, self.V, self.A, self.E, self.N, self.psi, self.kappa, self.theta, self.phi, self.edge_L, self.edge_R) = [np.asarray(r, dtype=np.float64) for r in data12]
        self.n = len(self.t)

    def _seg(self, arr, x):                  # clamp(searchsortedfirst(arr, x) - 1, 1, n-1), 0-based
        i = int(np.searchsorted(arr, x, side="left"))          # number of elements < x  == searchsortedfirst - 1
        return min(max(i, 1), self.n - 1) - 1

    def interp_by_s(self, s):                # (3P) Gridded(Linear()) + extrapolate(Line()): trajectories.jl:32-35
        j = min(max(int(np.searchsorted(self.s, s, side="right")), 1), self.n - 1) - 1
        w = (s - self.s[j]) / (self.s[j + 1] - self.s[j])
        f = lambda c: c[j] + w * (c[j + 1] - c[j])
        return dict(E=f(self.E), N=f(self.N), psi=f(self.psi), kappa=f(self.kappa), edge_L=f(self.edge_L), edge_R=f(self.edge_R))

    def at_time(self, t):                    # traj(t): :47-54
        i = self._seg(self.t, t)
        A = (self.V[i + 1] - self.V[i]) / (self.t[i + 1] - self.t[i])
        dt = t - self.t[i]
        return dict(s=self.s[i] + self.V[i] * dt + A * dt * dt / 2, V=self.V[i] + A * dt, A=A)

    def at_s(self, s):                       # traj[s]: :55-68
        i = self._seg(self.s, s)
        A = (self.V[i + 1] - self.V[i]) / (self.t[i + 1] - self.t[i])
        ds = s - self.s[i]
        if abs(A) < 1e-3 or s > self.s[-1]:
            dt = ds / self.V[i]
        else:
            dt = (math.sqrt(2 * A * ds + self.V[i] ** 2) - self.V[i]) / A
        si = self.interp_by_s(s)
        return dict(t=self.t[i] + dt, s=s, V=self.V[i] + A * dt, A=A, psi=si["psi"], kappa=si["kappa"])

    def path_coordinates(self, E, N):        # :71-94, math.jl:4-9
        x = np.array([E, N])
        d2min, imin = math.inf, -1
        for i in range(self.n - 1):
            p0 = np.array([self.E[i], self.N[i]]); p1 = np.array([self.E[i + 1], self.N[i + 1]])
            v = p1 - p0
            lam = min(max(v @ (x - p0) / (v @ v), 0), 1)
            p = (1 - lam) * p0 + lam * p1
            d2 = (p - x) @ (p - x)
            if d2 < d2min:
                d2min, imin = d2, i
        i = imin
        v = np.array([self.E[i + 1] - self.E[i], self.N[i + 1] - self.N[i]]); w = x - np.array([self.E[i], self.N[i]])
        ds = math.sqrt(max(w @ w - d2min, 0.0))
        s = self.s[i] + ds
        e = math.sqrt(d2min) * sign(v[0] * w[1] - v[1] * w[0])
        A = (self.V[i + 1] - self.V[i]) / (self.t[i + 1] - self.t[i])
        dt = ds / self.V[i] if abs(A) < 1e-3 else (math.sqrt(2 * A * ds + self.V[i] ** 2) - self.V[i]) / A
        return s, e, self.t[i] + dt


# ---------------------------------------------------------------------------------------------------------------------
# Julia's floating-point RANGES (Base twiceprecision.jl / range.jl / broadcast.jl of Julia 1.0, restated from memory: no Julia here, Base is not under /root/reference --
# a reading that could not be executed).  `x*(a:b)` is range(x*a, step=x, length=...): a StepRangeLen whose reference and step are TwicePrecision numbers, lifted to the
# exact rational where start and step have one (0.01 = 1/100, 0.2 = 1/5); `t .+ range` adds t to the reference in twice precision and stays a range; element i is ONE
# rounding of ref + (i - offset) step.  Written independently of oracle/julia_range.hpp: the error-free product is taken from exact rationals here, from fma there.
from fractions import Fraction as _Fr


def _canon2(big, little):
    h = big + little
    return h, (big - h) + little


def _add12(x, y):
    if abs(y) > abs(x):
        x, y = y, x
    return _canon2(x, y)


def _mul12(x, y):
    h = x * y
    if h == 0.0 or not math.isfinite(h):
        return h, h
    return _canon2(h, float(_Fr(x) * _Fr(y) - _Fr(h)))          # (the residual of a product is exactly representable)


def _truncbits(x, nb):
    if nb <= 0:
        return x
    u = np.float64(x).view(np.uint64)
    return float((u & np.uint64((0xFFFFFFFFFFFFFFFF << nb) & 0xFFFFFFFFFFFFFFFF)).view(np.float64))


def _tp_int(i):
    hi = _truncbits(float(i), 27)
    return _canon2(hi, float(i - int(hi)))


def _tp_div(x, y):
    hi = x[0] / y[0]
    uh, ul = _mul12(hi, y[0])
    lo = ((((x[0] - uh) - ul) + x[1]) - hi * y[1]) / y[0]
    return _canon2(hi, lo)


def _tp_trunc(v, nb):
    hi = _truncbits(v[0], nb)
    return hi, (v[0] - hi) + v[1]


def _tp_add(x, y):
    s_hi, s_lo = _add12(x[0], y)
    return _canon2(s_hi, s_lo + x[1])


def _rat(x):
    y = x; a = d = 1; b = c = 0; m = 16777216
    while abs(y) <= m:
        f = int(math.trunc(y)); y -= f
        a, c = f * a + c, a
        b, d = f * b + d, b
        if not max(abs(a), abs(b)) <= m:
            return c, d
        if b != 0 and float(a) / float(b) == x:
            break
        y = 1.0 / y if y != 0.0 else math.inf
    return a, b


def _nbitslen(length, offset):
    return 0 if length < 2 else min(27, int(math.ceil(math.log2(max(offset - 1, length - offset)))) + 1)


class JuliaRange:
    """StepRangeLen{Float64, TwicePrecision, TwicePrecision}: ref, step = (hi, lo) pairs; 1-based getindex."""

    def __init__(self, ref, step, length, offset):
        self.ref, self.step, self.len, self.offset = ref, step, length, offset

    def __getitem__(self, i):
        u = float(i - self.offset)
        shift_hi, shift_lo = u * self.step[0], u * self.step[1]
        x_hi, x_lo = _add12(self.ref[0], shift_hi)
        return x_hi + (x_lo + (shift_lo + self.ref[1]))

    def plus(self, x):
        """x .+ r (broadcast.jl: the result is again a range)"""
        return JuliaRange(_tp_add(self.ref, x), self.step, self.len, self.offset)

    def values(self):
        return np.array([self[i] for i in range(1, self.len + 1)])


def _floatrange(start_n, step_n, length, den):
    if length < 2 or step_n == 0:
        return JuliaRange(_tp_div(_tp_int(start_n), _tp_int(den)), _tp_div(_tp_int(step_n), _tp_int(den)), length, 1)
    imin = min(max(round(-start_n / step_n + 1), 1), length)
    ref_n = start_n + (imin - 1) * step_n
    return JuliaRange(_tp_div(_tp_int(ref_n), _tp_int(den)), _tp_trunc(_tp_div(_tp_int(step_n), _tp_int(den)), _nbitslen(length, imin)), length, imin)


def julia_range(a, st, length):
    """range(a, step = st, length = length) for Float64"""
    sn, sd = _rat(a); tn, td = _rat(st)
    if sd != 0 and td != 0 and sn / sd == a and tn / td == st:
        den = sd * td // math.gcd(sd, td)
        if abs(den * a) <= 2.0 ** 53 and abs(den * st) <= 2.0 ** 53 and den % sd == 0 and den % td == 0:
            return _floatrange(round(den * a), round(den * st), length, den)
    return JuliaRange((a, 0.0), (st, 0.0), length, 1)


def julia_scalar_times_unitrange(x, first, last):
    """x*(first:last)"""
    return julia_range(x * float(first), x * 1.0, max(last - first + 1, 0))


def julia_colon(start, step, stop):
    """start:step:stop for Float64"""
    between = lambda a, x, b: a <= x <= b or b <= x <= a
    tn, td = _rat(step)
    if td != 0 and tn / td == step:
        sn, sd = _rat(start); en, ed = _rat(stop)
        if sd != 0 and ed != 0 and sn / sd == start and en / ed == stop:
            den = sd * td // math.gcd(sd, td)
            if den != 0 and abs(start * den) <= 2.0 ** 53 and abs(step * den) <= 2.0 ** 53 and den % sd == 0 and den % td == 0:
                start_n, step_n = round(start * den), round(step * den)
                num, dd = den * en - ed * start_n + step_n * ed, step_n * ed
                length = max(0, abs(num) // abs(dd) * (1 if (num >= 0) == (dd >= 0) else -1))          # Julia's div truncates
                if between(start, start + (length - 1) * step, stop + step / 2) and not between(start, start + length * step, stop):
                    return _floatrange(start_n, step_n, length, den)
    lf = (stop - start) / step
    if lf < 0:
        length = 0
    elif lf == 0:
        length = 1
    else:
        length = round(lf) + 1
        stop2 = start + (length - 1) * step
        length -= int(start < stop < stop2) + int(start > stop > stop2)
    return JuliaRange((start, 0.0), (step, 0.0), length, 1)


def compute_time_steps(t0, N_short=10, N_long=20, dt_short=0.01, dt_long=0.2, use_correction_step=True, naive=False):
    """model_predictive_control.jl:17-30.  naive = True: the two-rounding form `t0 + dt*i` of rounds 1-5 (kept for A/B); default: Julia's range arithmetic (:25-26)."""
    t0_long = t0 + N_short * dt_short
    if use_correction_step:
        t0_long = dt_long * math.ceil((t0_long + dt_short) / dt_long - 1)
    if naive:
        ts = np.concatenate([t0 + dt_short * np.arange(N_short + 1), t0_long + dt_long * np.arange(1, N_long + 1)])
    else:
        ts = np.concatenate([julia_scalar_times_unitrange(dt_short, 0, N_short).plus(t0).values(), julia_scalar_times_unitrange(dt_long, 1, N_long).plus(t0_long).values()])
    return ts, np.diff(ts)


def simulate_times(dt, t_end, steps, t_start=0.0, naive=False):
    """The values `t` takes in `for t in 0:dt:trajectory.t[end]` (model_predictive_control.jl:87), shifted by t_start (the batch generalisation: t_start .+ (0:dt:t_end));
    naive = True: the accumulation t += dt of rounds 1-5."""
    if naive:
        out = [t_start]
        for _ in range(steps - 1):
            out.append(out[-1] + dt)
        return np.array(out)
    r = julia_colon(0.0, dt, t_end).plus(t_start)
    return np.array([r[k + 1] for k in range(steps)])


def compute_linearization_nodes(P, U, traj, state6, control3, ts, dt, N_short, N_long, time_offset=float("nan"), prev=None):
    """coupled_lat_long.jl:62-142.  state6 = (E, N, psi, Ux, Uy, r), control3 = (delta, Fxf, Fxr).  prev = None (cold) or
    (prev_ts, q_prev [N+1,6], u_prev_normalised [N+1,2], u_normalization) for the warm branch (:82-102, :189-195)."""
    E, Nn, psi, Ux0, Uy0, r0 = [float(v) for v in state6]
    d0, Fxf0, Fxr0 = [float(v) for v in control3]
    NN = N_short + N_long + 1
    s0, e0, _ = traj.path_coordinates(E, Nn)
    tj = traj.at_s(s0)
    ds = s0 - traj.at_time(ts[0])["s"]
    dpsi = adiff(psi, tj["psi"])
    qs = np.zeros((NN, 6)); us = np.zeros((NN, 2)); ps = np.zeros((NN, 4))
    qs[0] = [ds, Ux0, Uy0, r0, dpsi, e0]; us[0] = [d0, Fxf0 + Fxr0]; ps[0] = [tj["V"], tj["kappa"], 0, 0]
    if prev is not None:
        pts, qp_, up_, un = prev
        for i in range(1, NN):
            t = ts[i]
            tq = t if t < pts[-1] else pts[-1]
            j = min(max(int(np.searchsorted(pts, tq, side="right")), 1), NN - 1) - 1          # (3P) Gridded(Linear()) on the knots prev_ts
            w = (tq - pts[j]) / (pts[j + 1] - pts[j])
            q = (1 - w) * qp_[j] + w * qp_[j + 1]
            u = ((1 - w) * up_[j] + w * up_[j + 1]) * un
            s = traj.at_time(t)["s"] + q[0]
            tj = traj.at_s(s)
            qs[i] = q; us[i] = u; ps[i] = [tj["V"], tj["kappa"], 0, 0]
        return qs, us, ps
    s = s0
    sdp, cdp = math.sin(dpsi), math.cos(dpsi)
    V = Ux0 * cdp - Uy0 * sdp
    beta0 = math.atan2(Uy0, Ux0)
    sd, cd = math.sin(d0), math.cos(d0)
    Fyf0, _ = lateral_tire_forces(P, math.atan2(Uy0 + P["a"] * r0, Ux0) - d0, math.atan2(Uy0 - P["b"] * r0, Ux0), Fxf0, Fxr0, sd, cd)    # :110 (raw control, no limits)
    traj_mode = not math.isnan(time_offset)
    for i in range(NN):                      # i is the reference's i - 1
        tau = dt[i - 1] if i == NN - 1 else dt[i]
        tj = traj.at_s(s)
        ds = s - traj.at_time(ts[i])["s"]
        A_des = tj["A"] + U["k_V"] * (tj["V"] - V) / tau + (-U["k_s"] * ds / tau / tau if traj_mode else 0.0)
        A_des = jl_min(jl_max(A_des, (U["V_min"] - V) / tau), (U["V_max"] - V) / tau)
        if i == 0:
            qd = world_vehicle_model(P, [E, Nn, psi, Ux0, Uy0, r0], [d0, Fxf0 + Fxr0])             # :117
            A = (qd[3] - r0 * Uy0) * cdp - (qd[4] + r0 * Ux0) * sdp
        elif i <= N_short:
            est = steady_state_estimates(P, V, A_des, tj["kappa"], num_iters=1, r=r0, beta0=beta0, delta0=d0, Fyf0=Fyf0)
            qs[i] = [ds, Ux0, Uy0, r0, adiff(psi, tj["psi"]), e0]
            us[i] = [est["delta"], est["Fxf"] + est["Fxr"]]; ps[i] = [tj["V"], tj["kappa"], 0, 0]
            A = est["A"]
        else:
            est = steady_state_estimates(P, V, A_des, tj["kappa"])
            qs[i] = [ds, est["Ux"], est["Uy"], est["r"], -est["beta"], 0.0]
            us[i] = [est["delta"], est["Fxf"] + est["Fxr"]]; ps[i] = [tj["V"], tj["kappa"], 0, 0]
            A = est["A"]
        if i == NN - 1:
            break
        V = V + A * tau
        s = s + V * tau + A * tau * tau / 2
    return qs, us, ps


# ---------------------------------------------------------------------------------------------------------------------
def propagate(P, q, u0, p0, uf, pf, dt, ramp, nsub=10):
    """(3P) DifferentialDynamicsModels.propagate for generic dynamics: classical RK4, nsub sub-steps, control (and road parameters, which the
    reference concatenates into the control vector: coupled_lat_long.jl:336,348) held (StepControl) or interpolated linearly (RampControl)."""
    h = dt / nsub
    x = list(q)

    def f(xx, tau):
        w = tau / dt if ramp else 0.0
        u = [u0[k] + (uf[k] - u0[k]) * w for k in range(2)]
        p = [p0[k] + (pf[k] - p0[k]) * w for k in range(2)]
        return tracking_vehicle_model(P, xx, u, p)
    for i in range(nsub):
        t0 = i * h
        k1 = f(x, t0)
        k2 = f([x[k] + k1[k] * (h / 2) for k in range(6)], t0 + h / 2)
        k3 = f([x[k] + k2[k] * (h / 2) for k in range(6)], t0 + h / 2)
        k4 = f([x[k] + k3[k] * h for k in range(6)], t0 + h)
        x = [x[k] + (k1[k] + 2 * k2[k] + 2 * k3[k] + k4[k]) * (h / 6) for k in range(6)]
    return x


def linearize(P, q, u0, p0, uf, pf, dt, ramp, nsub=10):
    """(3P) LinearDynamicsModels.linearize(f, x, StepControl / RampControl; keep_control_dims = (1, 2)): A = dPhi/dx, B0 = dPhi/du0[keep],
    Bf = dPhi/duf[keep], c = Phi - A x - B0 u0 - Bf uf, derivatives by forward mode THROUGH the integrator."""
    n = 10
    e = np.eye(n)
    qd = [Dual(q[k], e[k]) for k in range(6)]
    u0d = [Dual(u0[k], e[6 + k]) for k in range(2)]
    ufd = [Dual(uf[k], e[8 + k]) for k in range(2)] if ramp else u0d
    phi = propagate(P, qd, u0d, list(p0), ufd, list(pf), dt, ramp, nsub)
    phi = [Dual.lift(v, n) for v in phi]
    J = np.array([v.g for v in phi]); val_ = np.array([v.v for v in phi])
    A = J[:, :6]; B0 = J[:, 6:8]; Bf = J[:, 8:10] if ramp else np.zeros((6, 2))
    c = val_ - A @ np.asarray(q) - B0 @ np.asarray(u0) - (Bf @ np.asarray(uf) if ramp else 0.0)
    return A, B0, Bf, c


def update_qp(P, U, qs, us, ps, dt, N_short, N_long, nsub=10, M_hji=(0.0, 0.0), b_hji=1.0):
    """update_QP!: coupled_lat_long.jl:315-368.  Returns a dict of the refreshed parameters; `flat` = the layout pg_get_qp documents
    (A[N][36] B0[N][12] Bf[N][12] c[N][6] H[N][8] G[N][4] dmin dmax fxmax ddmin ddmax dt [N each] q_curr[6] u_curr[2] M[2] b)."""
    N = N_short + N_long
    un = np.array([P["delta_max"], max(-P["Fx_min"], P["Fx_max"])])           # :199
    A = np.zeros((N, 6, 6)); B0 = np.zeros((N, 6, 2)); Bf = np.zeros((N, 6, 2)); c = np.zeros((N, 6))
    for t in range(N):
        ramp = t >= N_short
        At, B0t, Bft, ct = linearize(P, qs[t], us[t], ps[t][:2], us[t + 1], ps[t + 1][:2], dt[t], ramp, nsub)
        A[t] = At; B0[t] = B0t * un; Bf[t] = Bft * un; c[t] = ct                 # :338,350-351
    H = np.zeros((N, 4, 2)); G = np.zeros((N, 4)); dmin = np.zeros(N); dmax = np.zeros(N); fxmax = np.zeros(N); ddmin = np.zeros(N); ddmax = np.zeros(N)
    for t in range(N):
        Uxt = qs[t + 1][1]
        Fxf, Fxr = longitudinal_tire_forces(P, us[t + 1][1])
        d_lo, d_hi, Ht, Gt = stable_limits(P, Uxt, Fxf, Fxr)
        H[t] = Ht; G[t] = Gt
        dmin[t] = jl_max(d_lo, -P["delta_max"]) / un[0]; dmax[t] = jl_min(d_hi, P["delta_max"]) / un[0]
        fxmax[t] = jl_min(P["Px_max"] / Uxt, P["Fx_max"]) / un[1]
        ddmin[t] = -U["deltadot_max"] * dt[t] / un[0]; ddmax[t] = U["deltadot_max"] * dt[t] / un[0]
    out = dict(A=A, B0=B0, Bf=Bf, c=c, H=H, G=G, dmin=dmin, dmax=dmax, fxmax=fxmax, ddmin=ddmin, ddmax=ddmax, dt=np.asarray(dt), q_curr=np.asarray(qs[0]),
               u_curr=np.asarray(us[0]) / un, M=np.asarray(M_hji) * un, b=float(b_hji), un=un)
    out["flat"] = np.concatenate([A.ravel(), B0.ravel(), Bf.ravel(), c.ravel(), H.ravel(), G.ravel(), dmin, dmax, fxmax, ddmin, ddmax, out["dt"], out["q_curr"], out["u_curr"], out["M"], [out["b"]]])
    return out


# ---------------------------------------------------------------------------------------------------------------------
def assemble_canonical_qp(P, U, D, N_short, N_long):
    """construct_coupled_tracking_QP: coupled_lat_long.jl:197-313 in OSQP's form  min 1/2 x'Px + q'x  s.t.  l <= A x <= u  (dense A).
    Variable order = creation order (:233-239): q (6 x N+1, column-major), u (2 x N+1), sigma (2 x N), sigma_HJI (N_short), d_delta (N), d_Fx (N).
    Row order = @constraint statement order (:240-290).  Rows are written as  lhs - rhs  of each statement."""
    N = N_short + N_long; NN = N + 1
    oq, ou = 0, 6 * NN
    osg = ou + 2 * NN; osh = osg + 2 * N; odd = osh + N_short; odf = odd + N; n = odf + N
    vq = lambda i, t: oq + 6 * t + i
    vu = lambda i, t: ou + 2 * t + i
    vs = lambda i, t: osg + 2 * t + i
    rows, lo, hi = [], [], []
    INF = 1e30

    def add(coefs, l, u):
        r = np.zeros(n)
        for j, c in coefs:
            r[j] += c
        rows.append(r); lo.append(l); hi.append(u)
    for t in range(N):                                   # vec(sigma) >= 0
        for i in range(2):
            add([(vs(i, t), 1.0)], 0.0, INF)
    for t in range(N_short):                             # sigma_HJI >= 0
        add([(osh + t, 1.0)], 0.0, INF)
    for t in range(N):                                   # diff(delta) == d_delta
        add([(vu(0, t + 1), 1.0), (vu(0, t), -1.0), (odd + t, -1.0)], 0.0, 0.0)
    for t in range(N):                                   # diff(Fx) == d_Fx
        add([(vu(1, t + 1), 1.0), (vu(1, t), -1.0), (odf + t, -1.0)], 0.0, 0.0)
    for t in range(NN):
        add([(vq(1, t), 1.0)], U["V_min"], INF)
    for t in range(NN):
        add([(vq(1, t), 1.0)], -INF, U["V_max"])
    for t in range(NN):
        add([(vu(1, t), 1.0)], P["Fx_min"] / D["un"][1], INF)
    for i in range(6):
        add([(vq(i, 0), 1.0)], D["q_curr"][i], D["q_curr"][i])
    for i in range(2):
        add([(vu(i, 0), 1.0)], D["u_curr"][i], D["u_curr"][i])
    for t in range(N_short):                             # A q_t + B u_t + c == q_{t+1}
        for i in range(6):
            add([(vq(j, t), D["A"][t][i, j]) for j in range(6)] + [(vu(j, t), D["B0"][t][i, j]) for j in range(2)] + [(vq(i, t + 1), -1.0)], -D["c"][t][i], -D["c"][t][i])
    for t in range(N_short):                             # M u_t + b >= -sigma_HJI_t
        add([(vu(0, t), D["M"][0]), (vu(1, t), D["M"][1]), (osh + t, 1.0)], -D["b"], INF)
    for t in range(N_short, N):
        for i in range(6):
            add([(vq(j, t), D["A"][t][i, j]) for j in range(6)] + [(vu(j, t), D["B0"][t][i, j]) for j in range(2)] + [(vu(j, t + 1), D["Bf"][t][i, j]) for j in range(2)]
                + [(vq(i, t + 1), -1.0)], -D["c"][t][i], -D["c"][t][i])
    for t in range(N):
        add([(vu(0, t + 1), 1.0)], -INF, D["dmax"][t])
        add([(vu(0, t + 1), 1.0)], D["dmin"][t], INF)
        add([(vu(1, t + 1), 1.0)], -INF, D["fxmax"][t])
        for r in range(4):                               # H [Uy; r] - G <= sigma_t
            add([(vq(2, t + 1), D["H"][t][r, 0]), (vq(3, t + 1), D["H"][t][r, 1]), (vs(r // 2, t), -1.0)], -INF, D["G"][t][r])
        add([(odd + t, 1.0)], -INF, D["ddmax"][t])
        add([(odd + t, 1.0)], D["ddmin"][t], INF)
    Pd = np.zeros(n); qv = np.zeros(n)
    dt = D["dt"]
    for t in range(N):                                   # objective :292-308 (x'Qx terms => P = 2 Q)
        Pd[vq(0, t + 1)] = 2 * U["Q_ds"] * dt[t]; Pd[vq(4, t + 1)] = 2 * U["Q_dpsi"] * dt[t]; Pd[vq(5, t + 1)] = 2 * U["Q_e"] * dt[t]
        Pd[vu(0, t + 1)] = 2 * U["R_delta"] * dt[t]; Pd[vu(1, t + 1)] = 2 * U["R_Fx"] * dt[t]
        Pd[odd + t] = 2 * U["R_ddelta"] / dt[t]; Pd[odf + t] = 2 * U["R_dFx"] / dt[t]
        qv[vs(0, t)] = U["W_beta"] * dt[t]; qv[vs(1, t)] = U["W_r"] * dt[t]
    for t in range(N_short):                             # :343 W_HJI on the first N_HJI nodes only
        qv[osh + t] = U["W_HJI"] if t < U["N_HJI"] else 0.0
    return dict(Pd=Pd, q=qv, A=np.array(rows), l=np.array(lo), u=np.array(hi))
