#!/usr/bin/env python3
"""Generate tests/golden/paths/*.npz (decoded channels) and tests/golden/raw/ (three of the reference's own test DATA files, byte for byte) --
run in the build container only.  Inputs are /root/reference/test/path/*.world and *.msg; the decoders are the product's own
(pigeon.jl_amd/trajectories.py: read_world, decode_path_msg).  Outputs are plain data, no code.  The GPU box never sees /root/reference."""
import os
import shutil
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_pkg  # noqa: E402

REF = "/root/reference/test/path"
OUT = os.path.join(ROOT, "tests", "golden", "paths")
RAW = os.path.join(ROOT, "tests", "golden", "raw")


def main():
    pkg = load_pkg()
    T = sys.modules["pigeon_jl_amd.trajectories"]
    os.makedirs(OUT, exist_ok=True); os.makedirs(RAW, exist_ok=True)
    for name in ["skidpadoval", "vail", "EastPaddock", "paddockoval", "flidpadoval", "westpaddock", "newskidpadoval"]:
        d = T.read_world(os.path.join(REF, name + ".world"))
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **{k: d[k] for k in T.WORLD_KEYS}, isOpen=d["isOpen"])
        m = T.decode_path_msg(open(os.path.join(REF, name + ".msg"), "rb").read())      # cross-check against the ROS-serialised twin of the same path
        for k in T.WORLD_KEYS:
            assert np.allclose(m[k], d[k], rtol=0, atol=1e-9), (name, k)
    m = T.decode_path_msg(open(os.path.join(REF, "variable_speed.msg"), "rb").read())
    np.savez_compressed(os.path.join(OUT, "variable_speed.npz"), **{k: m[k] for k in T.WORLD_KEYS}, isOpen=m["isOpen"])
    for f in ["curvy.world", "curvy.msg", "variable_speed.msg"]:                           # raw data files for the ingest tests
        shutil.copy(os.path.join(REF, f), os.path.join(RAW, f))
    print("wrote", sorted(os.listdir(OUT)), sorted(os.listdir(RAW)))


if __name__ == "__main__":
    sys.exit(main())
