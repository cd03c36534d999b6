#!/usr/bin/env python3
"""Closed loop on the device (pg_simulate_dev, 4096 controllers, 40 steps behind 4): ms per step for the library selected by PIGEON_HIP_LIB.  usage: tools/gpu_closed_loop_ab.py [path]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import _load_pkg
import torch
pkg = _load_pkg()
path = sys.argv[1] if len(sys.argv) > 1 else "skidpadoval"
traj = pkg.load_path_fixture(path); B = 4096
state, control, t0, toff = pkg.synthetic.config2_inputs(traj, B, seed=12345)
for rep in range(3):
    m = pkg.BatchedTrajectoryTrackingMPC(traj, B, phase_timing=False)
    m.set_inputs(state, control, t0, time_offset=toff)
    m.simulate_(4); torch.cuda.synchronize(); t = time.perf_counter()
    m.simulate_(40); torch.cuda.synchronize(); t = time.perf_counter() - t
    st, it, _, _ = m.solve_info(); pol = m.polish_info()
    print(f"{os.environ.get('PIGEON_HIP_LIB', 'shipped')[-28:]} {path}: {1e3 * t / 40:.4f} ms/step = {B * 40 / t / 1e6:.3f} M solves/s; last step: solved {int(pkg.is_solved(st).sum())}/{B}, rounds mean {np.where(pol > 0, pol, 0).mean():.3f} max {pol.max()}, ipm {int((it > 0).sum())}", flush=True)
    m.close()
