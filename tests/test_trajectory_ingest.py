"""CPU tests of the trajectory ingest: `.world` files and serialised ROS `path` messages -> TrajectoryTube
(ros_integration.jl:13-19, math.jl:2, test/path/world2pathmsg.py).  The raw files under tests/golden/raw are the reference's own test DATA."""
import os

import numpy as np
import pytest

from conftest import ROOT

RAW = os.path.join(ROOT, "tests", "golden", "raw")


def test_world_file_and_its_ros_twin_decode_to_the_same_tube(pkg):
    """curvy.world and curvy.msg are the same path in the two formats the reference uses (the .msg files were produced from the .world files by
    test/path/world2pathmsg.py): the two decoders must agree channel by channel."""
    a = pkg.TrajectoryTube.from_world(os.path.join(RAW, "curvy.world"))
    b = pkg.TrajectoryTube.from_path_msg(os.path.join(RAW, "curvy.msg"))
    assert len(a) == len(b) == 1000
    assert np.max(np.abs(a.data - b.data)) < 1e-9
    c = pkg.TrajectoryTube.from_path_msg(open(os.path.join(RAW, "curvy.msg"), "rb").read())      # bytes, as the ROS callback receives them
    assert np.array_equal(b.data, c.data)


def test_message_only_path_matches_its_fixture(pkg):
    a = pkg.TrajectoryTube.from_path_msg(os.path.join(RAW, "variable_speed.msg"))
    b = pkg.load_path_fixture("variable_speed")
    assert np.array_equal(a.data, b.data)


def test_time_axis_is_the_inverse_cumulative_trapezoid(pkg):
    """TrajectoryTube(p::path): t = invcumtrapz(Ux_des, s) (ros_integration.jl:14, math.jl:2): dt_i = 2 ds_i / (V_i + V_{i+1}), t_1 = 0; phi = 0."""
    T = pkg.TrajectoryTube.from_world(os.path.join(RAW, "curvy.world"))
    assert T.t[0] == 0.0 and np.all(np.diff(T.t) > 0)
    assert np.allclose(np.diff(T.t), 2 * np.diff(T.s) / (T.V[:-1] + T.V[1:]), rtol=1e-9, atol=1e-12)
    assert np.all(T.phi == 0.0)
    raw = __import__("pigeon_jl_amd.trajectories", fromlist=["x"]).read_world(os.path.join(RAW, "curvy.world"))
    assert np.array_equal(T.edge_L, raw["edgeL_m"]) and np.array_equal(T.theta, raw["grade_rad"])


def test_malformed_inputs_are_rejected(pkg, tmp_path):
    T = __import__("pigeon_jl_amd.trajectories", fromlist=["x"])
    raw = open(os.path.join(RAW, "variable_speed.msg"), "rb").read()
    with pytest.raises(ValueError):
        T.decode_path_msg(raw[:-9])                 # truncated
    with pytest.raises(ValueError):
        T.decode_path_msg(raw + b"\\0" * 4)          # trailing bytes
    bad = tmp_path / "bad.world"
    bad.write_text("s_m: 0.0, 1.0\\nposE_m: 0.0, 1.0\\n")
    with pytest.raises(ValueError):
        T.read_world(str(bad))
    ragged = tmp_path / "ragged.world"
    ragged.write_text("\\n".join(f"{k}: 0.0, 1.0" + (", 2.0" if k == "posN_m" else "") for k in T.WORLD_KEYS) + "\\nisOpen: 1\\n")
    with pytest.raises(ValueError):
        T.read_world(str(ragged))
