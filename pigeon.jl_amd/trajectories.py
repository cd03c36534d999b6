"""TrajectoryTube host mirror — /root/reference/src/trajectories.jl:8-44, math.jl:1-2, ros_integration.jl:13-19."""
import os

import numpy as np

FIELDS = ["t", "s", "V", "A", "E", "N", "psi", "kappa", "theta", "phi", "edge_L", "edge_R"]


def invcumtrapz(y, x, x0=0.0):
    """math.jl:2"""
    return np.concatenate([[0.0], np.cumsum(2 * np.diff(x) / (y[:-1] + y[1:]))]) + x0


class TrajectoryTube:
    def __init__(self, t, s, V, A, E, N, psi, kappa, theta=None, phi=None, edge_L=None, edge_R=None):
        t = np.ascontiguousarray(t, dtype=np.float64)
        n = len(t)
        theta = np.zeros(n) if theta is None else theta          # trajectories.jl:43-44 defaults
        phi = np.zeros(n) if phi is None else phi
        edge_L = np.full(n, 4.0) if edge_L is None else edge_L
        edge_R = np.full(n, -4.0) if edge_R is None else edge_R
        cols = [t, s, V, A, E, N, psi, kappa, theta, phi, edge_L, edge_R]
        self.data = np.stack([np.ascontiguousarray(c, dtype=np.float64) for c in cols])
        assert self.data.shape == (12, n)                          # trajectories.jl:30-31

    def __len__(self):
        return self.data.shape[1]

    def __getattr__(self, name):
        if name in FIELDS:
            return self.data[FIELDS.index(name)]
        raise AttributeError(name)

    @classmethod
    def from_path(cls, p):
        """TrajectoryTube(p::path) of ros_integration.jl:13-16: t = invcumtrapz(Ux_des, s), phi = 0."""
        return cls(invcumtrapz(p["UxDes_mps"], p["s_m"]), p["s_m"], p["UxDes_mps"], p["AxDes_mps2"], p["posE_m"], p["posN_m"], p["psi_rad"], p["k_1pm"],
                   p["grade_rad"], 0 * p["grade_rad"], p["edgeL_m"], p["edgeR_m"])


def straight_trajectory(length, vel):
    """trajectories.jl:96-105"""
    return TrajectoryTube([0.0, length / vel], [0.0, length], [vel, vel], [0.0, 0.0], [0.0, 0.0], [0.0, length], [0.0, 0.0], [0.0, 0.0])


def load_path_fixture(name):
    """One of the reference's test paths (test/path/*.world), committed as data under tests/golden/paths/."""
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "paths")
    return TrajectoryTube.from_path(dict(np.load(os.path.join(root, name + ".npz"))))
