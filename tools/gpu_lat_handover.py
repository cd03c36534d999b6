#!/usr/bin/env python3
"""Round 6: the straggler hand-over of k_solve_lat (options lat_handover / lat_hand_target / lat_hand_min / lat_hand_cap) on the cold config-5 batch: solve phase, whole step,
and the answers against the single-launch solve (status, iteration counts, applied steering, working sets).
usage: tools/gpu_lat_handover.py [--walls] [--batch 4096] [--settings "0;1;2;1,lat_hand_target=2048;1,lat_hand_cap=14"]"""
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
    ap.add_argument("--walls", action="store_true"); ap.add_argument("--batch", type=int, default=4096); ap.add_argument("--reps", type=int, default=12)
    ap.add_argument("--settings", default="0;1"); ap.add_argument("--path", default="skidpadoval"); ap.add_argument("--Nl", type=int, default=40)
    a = ap.parse_args()
    import torch
    pkg = _load_pkg()
    B = a.batch
    traj = pkg.load_path_fixture(a.path)
    state, control, t0, toff = pkg.synthetic.config2_inputs(traj, B)
    ref = None
    for setting in a.settings.split(";"):
        parts = setting.split(",")
        opts = {"lat_handover": float(parts[0])}
        for p in parts[1:]:
            k, v = p.split("="); opts[k] = float(v)
        m = pkg.DecoupledTrajectoryTrackingMPC(pkg.X1(), traj, B, N_short=10, N_long=a.Nl, walls=a.walls, options=opts)
        m.set_stream(torch.cuda.current_stream().cuda_stream)
        m.set_inputs(state, control, t0, time_offset=toff)
        for _ in range(2):
            m.reset(); m.step_dev()
        torch.cuda.synchronize()
        ph = []
        for _ in range(a.reps):
            m.reset(); m.step_dev(); torch.cuda.synchronize(); ph.append(m.phase_ms())
        ph = np.array(ph)
        m.set_option("phase_timing", 0)
        t = time.perf_counter()
        for _ in range(a.reps):
            m.reset(); m.step_dev()
        torch.cuda.synchronize(); t = time.perf_counter() - t
        st, it, act, mu = m.solve_info(); pol = m.polish_info(); u = m.get_next_control(); x, sg = m.solution()
        line = (f"walls={int(a.walls)} {setting:28s}: nodes {np.median(ph[:, 0]):.3f} qp {np.median(ph[:, 1]):.3f} solve {np.median(ph[:, 2]):.3f} ms (min {ph[:, 2].min():.3f}, max {ph[:, 2].max():.3f})  step {1e3 * t / a.reps:.3f} ms = {B * a.reps / t / 1e6:.3f} M solves/s  "
                f"solved {int(pkg.is_solved(st).sum())}/{B} verified {int((pol >= 1).sum())} ipm iters mean {it.mean():.2f} max {it.max()} hand-over launches {m.get_option('stat_lat_handover_solves'):.0f} one-per-wavefront launches {m.get_option('stat_lat_one_per_wavefront_solves'):.0f}")
        if ref is None:
            ref = (st, it, act, u, pol, x)
        else:
            same = np.array_equal(st, ref[0]) and np.array_equal(it, ref[1]) and np.array_equal(act, ref[2])
            line += f"  | vs first: status/iters/sets identical {same}, status differs {int((st != ref[0]).sum())}, iters differ {int((it != ref[1]).sum())}, max |d2 - d2'| {np.max(np.abs(u[:, 0] - ref[3][:, 0])):.2e}, max |x - x'| {np.nanmax(np.abs(x - ref[5])):.2e}"
        print(line, flush=True)
        m.close()


if __name__ == "__main__":
    main()
