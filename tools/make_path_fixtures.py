#!/usr/bin/env python3
"""Generate tests/golden/paths/*.npz from the reference's own test DATA files (run in the build container only).

Inputs are /root/reference/test/path/*.world (YAML-ish "key: v1, v2, ..." lines; keys listed at
test/path/world2pathmsg.py:4-16) and variable_speed.msg (ROS wire format of safe_traffic_weaving/path, decoded
per SURVEY.md section 4).  Outputs are plain data: the raw channels, no code.  The GPU box never sees /root/reference.
"""
import os
import struct
import sys

import numpy as np

REF = "/root/reference/test/path"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "paths")
KEYS = ["s_m", "posE_m", "posN_m", "psi_rad", "k_1pm", "grade_rad", "edgeL_m", "edgeR_m", "UxDes_mps", "AxDes_mps2"]


def read_world(path):
    d = {}
    for line in open(path):
        if ":" not in line:
            continue
        k, v = line.split(":", 1)
        vals = [float(x) for x in v.replace("[", "").replace("]", "").split(",") if x.strip()]
        d[k.strip()] = np.array(vals)
    return d


def read_msg(path):
    b = open(path, "rb").read()
    o = 0
    seq, secs, nsecs, n = struct.unpack_from("<IIII", b, o); o += 16
    o += n          # frame_id
    o += 8          # one 8-byte scalar field (0)
    arrs = []
    for _ in range(10):
        (cnt,) = struct.unpack_from("<I", b, o); o += 4
        arrs.append(np.frombuffer(b, dtype="<f8", count=cnt, offset=o).copy()); o += 8 * cnt
    (is_open,) = struct.unpack_from("<q", b, o)
    d = dict(zip(KEYS, arrs))
    d["isOpen"] = np.array([float(is_open)])
    return d


def main():
    os.makedirs(OUT, exist_ok=True)
    for name in ["skidpadoval", "vail", "EastPaddock", "paddockoval", "flidpadoval", "westpaddock", "newskidpadoval"]:
        d = read_world(os.path.join(REF, name + ".world"))
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **{k: d[k] for k in KEYS}, isOpen=d["isOpen"])
        # cross-check against the ROS-serialised twin of the same path
        m = read_msg(os.path.join(REF, name + ".msg"))
        for k in KEYS:
            assert np.allclose(m[k], d[k], rtol=0, atol=1e-9), (name, k)
    m = read_msg(os.path.join(REF, "variable_speed.msg"))
    np.savez_compressed(os.path.join(OUT, "variable_speed.npz"), **m)
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    sys.exit(main())
