"""HJI value-grid files in the build's own flat format, and the contour helper of the RViz consumer.

The reference keeps its grid in deps/BicycleCAvoid.jld2 (downloaded by deps/build.jl:1-4; fields grid_knots, V_raw, gradV_raw:
src/HJI_computation.jl:48-57).  JLD2/HDF5 is not read here: tools/jld2_to_grid.jl (run once, on a machine with Julia) rewrites that file as

    bytes 0..7    magic  b"PGHJI\\x01\\x00\\x00"
    int32         ndim (= 7)
    int32[ndim]   dims                                  (grid_knots lengths)
    float32[sum]  knots, dimension after dimension      (grid_knots)
    float32[prod] V, column-major (dimension 1 fastest) (V_raw as Julia stores it)
    float32[7*prod] gradV, seven floats per node, same node order   (gradV_raw = reinterpret(Float32, coefs): src/HJI_computation.jl:61)

all little-endian; load_hji_grid returns exactly what BatchedTrajectoryTrackingMPC.set_hji_cache takes."""
import struct

import numpy as np

MAGIC = b"PGHJI\x01\x00\x00"


def save_hji_grid(path, knots, V, gradV):
    dims = [len(k) for k in knots]
    n = int(np.prod(dims, dtype=np.int64))
    V = np.ascontiguousarray(V, dtype="<f4").reshape(-1); g = np.ascontiguousarray(gradV, dtype="<f4").reshape(-1)
    if V.size != n or g.size != 7 * n or len(dims) != 7:
        raise ValueError("save_hji_grid: need 7 knot vectors, prod(dims) values and 7 x prod(dims) gradient entries")
    with open(path, "wb") as f:
        f.write(MAGIC); f.write(struct.pack("<i", 7)); f.write(struct.pack("<7i", *dims))
        for k in knots:
            f.write(np.ascontiguousarray(k, dtype="<f4").tobytes())
        f.write(V.tobytes()); f.write(g.tobytes())


def load_hji_grid(path):
    """(knots [7 float32 arrays], V [prod dims] float32 column-major, gradV [prod dims, 7] float32)"""
    with open(path, "rb") as f:
        if f.read(8) != MAGIC:
            raise ValueError(f"{path}: not an HJI grid file of this build (bad magic)")
        (nd,) = struct.unpack("<i", f.read(4))
        if nd != 7:
            raise ValueError(f"{path}: {nd}-dimensional grid (7 expected)")
        dims = struct.unpack("<7i", f.read(28))
        if min(dims) < 2:
            raise ValueError(f"{path}: every dimension needs at least two knots")
        n = int(np.prod(dims, dtype=np.int64))
        knots = [np.frombuffer(f.read(4 * d), dtype="<f4").copy() for d in dims]
        V = np.frombuffer(f.read(4 * n), dtype="<f4")
        g = np.frombuffer(f.read(28 * n), dtype="<f4")
        if V.size != n or g.size != 7 * n or f.read(1):
            raise ValueError(f"{path}: truncated or oversized grid file")
        if any(np.any(np.diff(k) <= 0) for k in knots):
            raise ValueError(f"{path}: knot vectors must be strictly increasing")
    return knots, V.copy(), g.reshape(n, 7).copy()


def trace_zero_contour(X, Y, cross_x, cross_y):
    """First zero-level line of one slice as an ordered vertex list [(x, y), ...] -- what update_HJI_contour_marker! publishes (rviz.jl:63-68: c.lines[1]).
    cross_x [n1-1, n2], cross_y [n1, n2-1] from hji_value_slice.  Marching squares over the cells, walking from edge to edge; saddle cells (four crossings)
    are split like Contour.jl's default (pairs in index order).  Returns [] when the slice has no crossing."""
    n1, n2 = len(X), len(Y)
    edges = {}
    for i in range(n1 - 1):
        for j in range(n2):
            if cross_x[i, j] == cross_x[i, j]:
                edges[("x", i, j)] = (float(cross_x[i, j]), float(Y[j]))
    for i in range(n1):
        for j in range(n2 - 1):
            if cross_y[i, j] == cross_y[i, j]:
                edges[("y", i, j)] = (float(X[i]), float(cross_y[i, j]))
    if not edges:
        return []
    link = {}
    for i in range(n1 - 1):
        for j in range(n2 - 1):
            e = [k for k in (("x", i, j), ("y", i + 1, j), ("x", i, j + 1), ("y", i, j)) if k in edges]
            for a, b in zip(e[0::2], e[1::2]):
                link.setdefault(a, []).append(b); link.setdefault(b, []).append(a)
    start = next((k for k in sorted(edges) if len(link.get(k, [])) == 1), sorted(edges)[0])
    line, seen, cur = [edges[start]], {start}, start
    while True:
        nxt = [k for k in link.get(cur, []) if k not in seen]
        if not nxt:
            break
        cur = nxt[0]; seen.add(cur); line.append(edges[cur])
    return line
