"""Diagnostic (round 6): is the lateral closed loop reproducible bit for bit?  Two handles, the same calls; prints the first step at which the recorded histories differ,
under a few option sets (which launch path is responsible).   usage: tools/gpu_lat_repro.py [--walls] [--steps 12]"""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import _load_pkg
pkg = _load_pkg()
ap = argparse.ArgumentParser(); ap.add_argument("--walls", action="store_true"); ap.add_argument("--steps", type=int, default=12); ap.add_argument("--batch", type=int, default=4096)
a = ap.parse_args()
traj = pkg.load_path_fixture("skidpadoval")
B = a.batch
state, control, t0, toff = pkg.synthetic.config2_inputs(traj, B, seed=12345)
import json
ap2 = None
base = {"lat_handover": 0, "lat_single_max": 0, "lat_split": 0}
sets = [dict(base, **json.loads(x)) for x in os.environ.get("PG_SETS", "").split(";") if x] or [{}, {"lat_split": 0}, {"lat_single_max": 0}, {"lat_handover": 0, "lat_single_max": 0}, base, {"nodes_serial": 1}]
for opts in sets:
    runs = []
    for rep in range(2):
        m = pkg.DecoupledTrajectoryTrackingMPC(pkg.X1(), traj, B, N_short=10, N_long=40, walls=a.walls, options=opts)
        m.set_inputs(state, control, t0, time_offset=toff)
        s, c, t, sh, ch = m.simulate_(a.steps, record=True)
        runs.append((np.asarray(sh), np.asarray(ch)))
        m.close()
    sh0, sh1 = runs[0][0], runs[1][0]; ch0, ch1 = runs[0][1], runs[1][1]
    dif_c = [int((ch0[k] != ch1[k]).any(axis=-1).sum()) for k in range(ch0.shape[0])]
    dif_s = [int((sh0[k] != sh1[k]).any(axis=-1).sum()) for k in range(sh0.shape[0])]
    print(opts, "controls differ per step:", dif_c, "states:", dif_s, "max |dc|", float(np.max(np.abs(ch0 - ch1))), flush=True)
