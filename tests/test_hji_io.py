"""CPU tests: the build's HJI grid file format (written by tools/jld2_to_grid.jl on the Julia side, read by pigeon.jl_amd/hji_io.py) and the contour tracer."""
import struct

import numpy as np
import pytest


def test_grid_file_round_trip_and_layout(pkg, tmp_path):
    knots, V, g = pkg.synthetic.hji_grid(dims=(5, 4, 3, 2, 2, 3, 2), seed=2)
    f = str(tmp_path / "grid.pghji")
    pkg.save_hji_grid(f, knots, V, g)
    k2, V2, g2 = pkg.load_hji_grid(f)
    assert all(np.array_equal(a, b) for a, b in zip(knots, k2)) and np.array_equal(V2, V) and np.array_equal(g2, g)
    raw = open(f, "rb").read()                       # the layout tools/jld2_to_grid.jl writes
    assert raw[:8] == b"PGHJI\x01\x00\x00" and struct.unpack_from("<i", raw, 8) == (7,) and struct.unpack_from("<7i", raw, 12) == (5, 4, 3, 2, 2, 3, 2)
    nk = 5 + 4 + 3 + 2 + 2 + 3 + 2; n = 5 * 4 * 3 * 2 * 2 * 3 * 2
    assert len(raw) == 40 + 4 * nk + 4 * n + 28 * n
    assert np.frombuffer(raw, "<f4", 5, 40)[1] == knots[0][1]
    assert np.frombuffer(raw, "<f4", 1, 40 + 4 * nk + 4 * 1)[0] == V[1]              # dimension 1 fastest
    assert np.array_equal(np.frombuffer(raw, "<f4", 7, 40 + 4 * nk + 4 * n + 28 * 3), g[3])


def test_grid_file_errors(pkg, tmp_path):
    knots, V, g = pkg.synthetic.hji_grid(dims=(3, 3, 2, 2, 2, 2, 2), seed=2)
    f = str(tmp_path / "grid.pghji")
    pkg.save_hji_grid(f, knots, V, g)
    raw = open(f, "rb").read()
    for bad in (b"XXXXX" + raw[5:], raw[:-4], raw + b"\0"):
        p = tmp_path / "bad.pghji"; p.write_bytes(bad)
        with pytest.raises(ValueError):
            pkg.load_hji_grid(str(p))
    with pytest.raises(ValueError):
        pkg.save_hji_grid(f, knots[:6], V, g)


def test_contour_tracer_orders_the_vertices_of_a_circle(pkg):
    X = np.linspace(-4, 4, 33); Y = np.linspace(-3, 3, 25)
    V = np.sqrt(X[:, None] ** 2 + Y[None, :] ** 2) - 2.0
    up = V > 0
    cx = np.where(up[:-1] != up[1:], X[:-1, None] + (0 - V[:-1]) / (V[1:] - V[:-1]) * (X[1:, None] - X[:-1, None]), np.nan)
    cy = np.where(up[:, :-1] != up[:, 1:], Y[None, :-1] + (0 - V[:, :-1]) / (V[:, 1:] - V[:, :-1]) * (Y[None, 1:] - Y[None, :-1]), np.nan)
    line = np.array(pkg.trace_zero_contour(X, Y, cx, cy))
    assert len(line) == np.isfinite(cx).sum() + np.isfinite(cy).sum()           # one closed line through every crossing
    assert np.max(np.abs(np.hypot(line[:, 0], line[:, 1]) - 2.0)) < 5e-3        # vertices sit on the level set (linear interpolation error)
    assert np.max(np.hypot(*np.diff(line, axis=0).T)) < 0.5                     # consecutive vertices are neighbours: the order is a walk along the line
    assert pkg.trace_zero_contour(X, Y, np.full_like(cx, np.nan), np.full_like(cy, np.nan)) == []
