"""Synthetic inputs of BASELINE.md section 3 (seeded, numpy default_rng / PCG64) and a synthetic 7-D HJI grid (the real
deps/BicycleCAvoid.jld2 of the reference is not in the repository: deps/build.jl:1-4)."""
import numpy as np


def path_pose(traj, s):
    """(E, N, psi, kappa, V, t) of the tube at arclength s by linear interpolation of the samples."""
    f = lambda c: np.interp(s, traj.s, c)
    return f(traj.E), f(traj.N), f(traj.psi), f(traj.kappa), f(traj.V), f(traj.t)


def config2_inputs(traj, B, seed=12345, traj_mode=True, s_range=None):
    """Config 2 of BASELINE.md: randomised x0 along one test path.  Returns state [B,6], control [B,3], t0 [B], time_offset [B].
    s_range overrides the arclength window (default [5, s_end - 60] m; short paths need their own)."""
    rng = np.random.default_rng(seed)
    lo, hi = (5.0, traj.s[-1] - 60.0) if s_range is None else s_range
    s = rng.uniform(lo, hi, B)
    E, N, psi, kappa, V, t = path_pose(traj, s)
    e = rng.uniform(-0.5, 0.5, B)
    # psi is measured from North (vehicle_dynamics.jl:127): heading (-sin psi, cos psi), left normal (-cos psi, -sin psi)
    state = np.stack([E - e * np.cos(psi), N - e * np.sin(psi), psi + rng.uniform(-0.1, 0.1, B), V * rng.uniform(0.9, 1.1, B), rng.uniform(-0.2, 0.2, B),
                      kappa * V + rng.uniform(-0.05, 0.05, B)], axis=1)
    delta0 = rng.uniform(-0.05, 0.05, B)
    Fx0 = rng.uniform(-500.0, 500.0, B)
    control = np.stack([delta0, np.where(Fx0 > 0, 0.0, 0.6) * Fx0, np.where(Fx0 > 0, 1.0, 0.4) * Fx0], axis=1)   # longitudinal_tire_forces
    t0 = t + rng.uniform(-0.2, 0.2, B)
    toff = np.zeros(B) if traj_mode else np.full(B, np.nan)
    return state, control, t0, toff


def other_cars(state, seed=777):
    """Config 3: other car within +-15 m x +-4 m of the ego (ego frame), heading within +-0.3 rad, speed U(2,10)."""
    rng = np.random.default_rng(seed)
    B = state.shape[0]
    dx = rng.uniform(-15, 15, B); dy = rng.uniform(-4, 4, B)
    psi = state[:, 2]
    # ego frame: forward = (-sin psi, cos psi), left = (-cos psi, -sin psi)
    E = state[:, 0] + dx * (-np.sin(psi)) + dy * (-np.cos(psi))
    N = state[:, 1] + dx * (np.cos(psi)) + dy * (-np.sin(psi))
    return np.stack([E, N, psi + rng.uniform(-0.3, 0.3, B), rng.uniform(2, 10, B)], axis=1)


def hji_grid(dims=(13, 13, 9, 9, 9, 9, 9), seed=3):
    """Non-uniform knots and an analytic signed-distance-like value function V(x) = sqrt(dE^2/4 + dN^2 + 1) - 3 + 0.05 Ux - 0.02 V_them + 0.1 cos(dpsi)
    so that gradV is known in closed form.  Returns (knots list, V [prod dims] column-major, gradV [prod dims, 7])."""
    rng = np.random.default_rng(seed)
    lo = np.array([-20.0, -8.0, -np.pi, 0.5, -2.0, 0.0, -1.0]); hi = np.array([20.0, 8.0, np.pi, 14.0, 2.0, 12.0, 1.0])
    knots = []
    for d in range(7):
        u = np.linspace(0, 1, dims[d])
        u[1:-1] += rng.uniform(-0.25, 0.25, dims[d] - 2) / (dims[d] - 1)
        knots.append((lo[d] + (hi[d] - lo[d]) * u).astype(np.float32))
    G = np.meshgrid(*[k.astype(np.float64) for k in knots], indexing="ij")
    r = np.sqrt(G[0] ** 2 / 4 + G[1] ** 2 + 1.0)
    V = r - 3.0 + 0.05 * G[3] - 0.02 * G[5] + 0.1 * np.cos(G[2])
    g = np.zeros(V.shape + (7,))
    g[..., 0] = G[0] / (4 * r); g[..., 1] = G[1] / r; g[..., 2] = -0.1 * np.sin(G[2]); g[..., 3] = 0.05; g[..., 5] = -0.02
    # Julia arrays are column-major (dim 1 fastest): flatten in Fortran order
    Vf = np.asarray(V, dtype=np.float32).reshape(-1, order="F")
    gf = np.stack([np.asarray(g[..., k], dtype=np.float32).reshape(-1, order="F") for k in range(7)], axis=1)
    return knots, Vf, gf


def hji_grid_large(dims=(13, 13, 9, 9, 9, 9, 9), seed=3):
    """BASELINE config 3 grid (~10 M nodes: V 40 MB + gradV 279 MB, larger than the 256 MiB Infinity Cache) built with float32
    broadcasting (same analytic V as hji_grid, a few seconds instead of minutes)."""
    rng = np.random.default_rng(seed)
    lo = np.array([-20.0, -8.0, -np.pi, 0.5, -2.0, 0.0, -1.0]); hi = np.array([20.0, 8.0, np.pi, 14.0, 2.0, 12.0, 1.0])
    knots = []
    for d in range(7):
        u = np.linspace(0, 1, dims[d])
        u[1:-1] += rng.uniform(-0.25, 0.25, dims[d] - 2) / (dims[d] - 1)
        knots.append((lo[d] + (hi[d] - lo[d]) * u).astype(np.float32))
    # Fortran order (dim 1 fastest) == C order of the reversed axes
    ax = [knots[d].astype(np.float32).reshape([-1 if k == d else 1 for k in range(6, -1, -1)]) for d in range(7)]   # axis position 6-d
    shape = tuple(dims[::-1])
    r = np.sqrt(ax[0] ** 2 / 4 + ax[1] ** 2 + 1.0).astype(np.float32)
    V = np.broadcast_to(r - 3.0 + 0.05 * ax[3] - 0.02 * ax[5] + 0.1 * np.cos(ax[2]), shape).astype(np.float32).reshape(-1)
    g = np.zeros((V.size, 7), dtype=np.float32)
    g[:, 0] = np.broadcast_to(ax[0] / (4 * r), shape).reshape(-1)
    g[:, 1] = np.broadcast_to(ax[1] / r, shape).reshape(-1)
    g[:, 2] = np.broadcast_to(-0.1 * np.sin(ax[2]), shape).reshape(-1)
    g[:, 3] = 0.05; g[:, 5] = -0.02
    return knots, V, g


def hji_queries(knots, n, seed=9, margin=0.0):
    """n relative states uniformly inside the grid (margin > 0 pushes a share of them out of bounds)."""
    rng = np.random.default_rng(seed)
    lo = np.array([k[0] for k in knots], dtype=np.float64); hi = np.array([k[-1] for k in knots], dtype=np.float64)
    return lo + (hi - lo) * rng.uniform(-margin, 1 + margin, (n, 7))
