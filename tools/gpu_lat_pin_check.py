#!/usr/bin/env python3
"""Lateral N = 50 batch, cold step and the warm step behind `burn` closed-loop steps, pinned inputs on / off (option lat_pin): every instance against the oracle's verified KKT point.
usage: tools/gpu_lat_pin_check.py [walls 0/1] [burn]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import _load_pkg
pkg = _load_pkg()
from oracle import oracle as oracle_mod
import scipy.sparse as sp
import test_gpu_decoupled as T
T.sp = sp
walls = bool(int(sys.argv[1])) if len(sys.argv) > 1 else True
burn = int(sys.argv[2]) if len(sys.argv) > 2 else 3
B, Ns, Nl = 4096, 10, 40
traj = pkg.load_path_fixture("skidpadoval")
state, control, t0, toff = pkg.synthetic.config2_inputs(traj, B)
for pin in [int(v) for v in (sys.argv[3].split(',') if len(sys.argv) > 3 else ['1', '0'])]:
    mpc = pkg.DecoupledTrajectoryTrackingMPC(pkg.X1(), traj, B, N_short=Ns, N_long=Nl, walls=walls, options={"lat_pin": pin})
    mpc.set_inputs(state, control, t0, time_offset=toff)
    for phase in ("cold", "warm"):
        if phase == "warm":
            s, c, t, _, _ = mpc.simulate_(burn)
            mpc.set_inputs(s, c, t, time_offset=toff)
        mpc.step_dev(); mpc.synchronize()
        st, it, act, mu = mpc.solve_info(); pol = mpc.polish_info()
        res = T.check_lateral_batch_against_oracle(pkg, oracle_mod, traj, mpc, B, Ns, Nl, walls)
        w = int(np.argmax(res[:, 0]))
        print(f"lat_pin={pin} walls={walls} {phase}: verified {int((pol >= 1).sum())}/{B} unverified {int((pol < 0).sum())} ipm-iters mean {it.mean():.2f} max {it.max()} | |d2-d2*| max {res[:, 0].max():.2e} (instance {w}, polish {pol[w]}, iters {it[w]}) "
              f">1e-7: {int((res[:, 0] > 1e-7).sum())} >1e-6: {int((res[:, 0] > 1e-6).sum())} | objective gap max {res[:, 1].max():.2e} | horizon max {res[:, 3].max():.2e} | phase ms {mpc.phase_ms()}", flush=True)
    mpc.close()
