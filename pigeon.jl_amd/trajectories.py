"""TrajectoryTube host mirror — /root/reference/src/trajectories.jl:8-44, math.jl:1-2, ros_integration.jl:13-19."""
import os

import numpy as np

FIELDS = ["t", "s", "V", "A", "E", "N", "psi", "kappa", "theta", "phi", "edge_L", "edge_R"]


WORLD_KEYS = ["s_m", "posE_m", "posN_m", "psi_rad", "k_1pm", "grade_rad", "edgeL_m", "edgeR_m", "UxDes_mps", "AxDes_mps2"]
# field order of the ROS message safe_traffic_weaving/path as serialised in the reference's test data (test/path/*.msg; the .world <-> message field
# names are listed at test/path/world2pathmsg.py:4-16): header, one 8-byte scalar, ten float64[] arrays in this order, then isOpen (int64)
PATH_MSG_ARRAYS = WORLD_KEYS


def read_world(path):
    """Decode a `.world` path file (YAML: one `key: v1, v2, ...` line per channel; test/path/world2pathmsg.py:18-25) into a dict of float64 arrays."""
    d = {}
    with open(path) as f:
        for line in f:
            if ":" not in line:
                continue
            k, v = line.split(":", 1)
            vals = [float(x) for x in v.replace("[", "").replace("]", "").replace("'", "").replace('"', "").split(",") if x.strip()]
            d[k.strip()] = np.array(vals, dtype=np.float64)
    missing = [k for k in WORLD_KEYS if k not in d]
    if missing:
        raise ValueError(f"{path}: not a .world path file (missing {missing})")
    n = len(d["s_m"])
    if any(len(d[k]) != n for k in WORLD_KEYS):
        raise ValueError(f"{path}: channels of different lengths")
    return d


def decode_path_msg(buf):
    """Decode the ROS wire format (little-endian) of a serialised `path` message -- what the ROS loop receives on its path topic
    (ros_integration.jl:13-19,44-47) -- into the same dict read_world returns.  Layout: std_msgs/Header (uint32 seq, uint32 secs, uint32 nsecs,
    uint32 len + frame_id bytes), one 8-byte scalar, ten (uint32 count + count x float64) arrays in PATH_MSG_ARRAYS order, int64 isOpen."""
    import struct
    b = bytes(buf)
    try:
        o = 0
        seq, secs, nsecs, n = struct.unpack_from("<IIII", b, o); o += 16
        frame_id = b[o:o + n].decode("ascii", "replace"); o += n
        o += 8
        d = {}
        for k in PATH_MSG_ARRAYS:
            (cnt,) = struct.unpack_from("<I", b, o); o += 4
            if o + 8 * cnt > len(b):
                raise ValueError("array runs past the end of the buffer")
            d[k] = np.frombuffer(b, dtype="<f8", count=cnt, offset=o).astype(np.float64); o += 8 * cnt
        (is_open,) = struct.unpack_from("<q", b, o); o += 8
    except struct.error as e:
        raise ValueError(f"truncated path message: {e}")
    if o != len(b):
        raise ValueError(f"{len(b) - o} trailing bytes after the path message")
    if len({len(d[k]) for k in PATH_MSG_ARRAYS}) != 1:
        raise ValueError("path message with channels of different lengths")
    d["isOpen"] = np.array([float(is_open)]); d["frame_id"] = frame_id
    return d


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


def _from_world(cls, path):
    """TrajectoryTube from a `.world` file (the format of the reference's test paths): read_world + the conversion of ros_integration.jl:13-16."""
    return cls.from_path(read_world(path))


def _from_path_msg(cls, msg):
    """TrajectoryTube from a serialised ROS `path` message (bytes, or the name of a file holding them): decode_path_msg + ros_integration.jl:13-16."""
    if isinstance(msg, (str, os.PathLike)):
        with open(msg, "rb") as f:
            msg = f.read()
    return cls.from_path(decode_path_msg(msg))


TrajectoryTube.from_world = classmethod(_from_world)
TrajectoryTube.from_path_msg = classmethod(_from_path_msg)


def straight_trajectory(length, vel):
    """trajectories.jl:96-105"""
    return TrajectoryTube([0.0, length / vel], [0.0, length], [vel, vel], [0.0, 0.0], [0.0, 0.0], [0.0, length], [0.0, 0.0], [0.0, 0.0])


def load_path_fixture(name):
    """One of the reference's test paths (test/path/*.world), committed as data under tests/golden/paths/."""
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "paths")
    return TrajectoryTube.from_path(dict(np.load(os.path.join(root, name + ".npz"))))
