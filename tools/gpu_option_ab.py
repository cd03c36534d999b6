#!/usr/bin/env python3
"""A/B of one build-defined option (pg_set_option) on cold steps of config 2: per-phase times (median of HIP-event triples over `reps` steps) and solves/s over a timed loop.
usage: tools/gpu_option_ab.py OPTION v1,v2,... [--paths skidpadoval,vail] [--batch 4096] [--precision f64] [--warm] [--formulation coupled|decoupled] [--walls]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import _load_pkg  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("option"); ap.add_argument("values")
    ap.add_argument("--paths", default="skidpadoval,vail"); ap.add_argument("--batch", type=int, default=4096); ap.add_argument("--precision", default="f64")
    ap.add_argument("--reps", type=int, default=30); ap.add_argument("--warm", action="store_true"); ap.add_argument("--formulation", default="coupled"); ap.add_argument("--walls", action="store_true")
    ap.add_argument("--rounds", type=int, default=2, help="alternate the values this many times (drift shows as a spread between rounds)")
    a = ap.parse_args()
    import torch
    pkg = _load_pkg()
    B = a.batch
    vals = [float(v) for v in a.values.split(",")]
    for path in a.paths.split(","):
        traj = pkg.load_path_fixture(path)
        kw = dict(s_range=(2.0, 0.4 * float(traj.s[-1]))) if float(traj.s[-1]) <= 100 else {}
        state, control, t0, toff = pkg.synthetic.config2_inputs(traj, B, seed=12345, **kw)
        for rnd in range(a.rounds):
            for v in vals:
                opts = {} if a.option == "none" else {a.option: v}
                if a.formulation == "coupled":
                    m = pkg.BatchedTrajectoryTrackingMPC(traj, B, precision=a.precision, options=opts)
                else:
                    m = pkg.DecoupledTrajectoryTrackingMPC(pkg.X1(), traj, B, N_short=10, N_long=40, precision=a.precision, walls=a.walls, options=opts)
                m.set_stream(torch.cuda.current_stream().cuda_stream)
                m.set_inputs(state, control, t0, time_offset=toff)
                for _ in range(3):
                    if not a.warm: m.reset()
                    m.step_dev()
                torch.cuda.synchronize()
                ph = []
                for _ in range(a.reps):
                    if not a.warm: m.reset()
                    m.step_dev(); torch.cuda.synchronize()
                    try: ph.append(m.phase_ms())
                    except Exception: ph.append([0.0, 0.0, 0.0])
                ph = np.median(np.array(ph), axis=0)
                t = time.perf_counter()
                for _ in range(a.reps):
                    if not a.warm: m.reset()
                    m.step_dev()
                torch.cuda.synchronize(); t = time.perf_counter() - t
                st, it, _, _ = m.solve_info(); pol = m.polish_info()
                print(f"{path:14s} {a.option}={v:g} round {rnd}: phases {ph[0]:.4f} {ph[1]:.4f} {ph[2]:.4f} ms  loop {1e3 * t / a.reps:.4f} ms/step = {B * a.reps / t / 1e6:.3f} M solves/s  "
                      f"solved {int(pkg.is_solved(st).sum())}/{B} ipm {int((it > 0).sum())} rounds mean {np.where(pol > 0, pol, 0).mean():.3f} max {pol.max()} unverified {int((pol < 0).sum())}", flush=True)
                m.close()


if __name__ == "__main__":
    main()
