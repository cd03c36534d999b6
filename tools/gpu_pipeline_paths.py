"""Pipelined nodes + update_QP launch (pg_set_pipeline) against the launch-per-phase sequence on all eight reference paths and at the batch sizes that bound its
use (2048, 4096, 16384; fp64): bit-identity of nodes, QP data and controls of a cold step, and the time per step either way.
    python tools/gpu_pipeline_paths.py"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge._load_pkg()
PATHS = ["skidpadoval", "vail", "EastPaddock", "variable_speed", "flidpadoval", "newskidpadoval", "paddockoval", "westpaddock"]
for name in PATHS:
    traj = pkg.load_path_fixture(name)
    for n in ((2048, 4096, 16384) if name == "skidpadoval" else (4096,)):
        sr = None if traj.s[-1] > 90 else (2.0, 0.4 * traj.s[-1])
        state, control, t0, toff = pkg.synthetic.config2_inputs(traj, n, seed=17, s_range=sr)
        out = {}; ms = {}
        for piped in (False, True):
            mpc = pkg.BatchedTrajectoryTrackingMPC(traj, n)
            mpc.set_pipeline(piped)
            mpc.set_inputs(state, control, t0, time_offset=toff)
            mpc.step_dev(); mpc.synchronize()
            out[piped] = [np.concatenate([a.reshape(n, -1) for a in mpc.nodes()], axis=1), mpc.qp_data().copy(), mpc.get_next_control().copy(), mpc.solve_info()[0].copy()]
            for _ in range(3):
                mpc.reset(); mpc.step_dev()
            mpc.synchronize(); t = time.perf_counter()
            for _ in range(20):
                mpc.reset(); mpc.step_dev()
            mpc.synchronize(); ms[piped] = (time.perf_counter() - t) / 20 * 1e3
            mpc.close()
        same = all(np.array_equal(a, b, equal_nan=True) for a, b in zip(out[False], out[True]))
        print(f"{name:16s} B={n:6d} identical={same} solved={int((out[True][3] == pkg.SOLVED).sum())}/{n}  ms per cold step: per-phase {ms[False]:.3f}  pipelined {ms[True]:.3f}  ({ms[False] / ms[True]:.3f}x)", flush=True)
        assert same
