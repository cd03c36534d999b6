"""Robustness of the clipped working-set guess (option "clip_guess") across the reference's test paths: cold step of 4096 instances per path, with and without it --
instances that end in the interior point, active-set rounds per instance, solve-phase time.  Usage (GPU box): python tools/gpu_clip_paths.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import _load_pkg
pkg = _load_pkg()
B = 4096
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "paths")
un = np.array([0.314159, 16793.7, 16793.7])
for name in sorted(f[:-4] for f in os.listdir(root) if f.endswith(".npz")):
    traj = pkg.load_path_fixture(name)
    kw = dict(s_range=(2.0, 0.4 * float(traj.s[-1]))) if float(traj.s[-1]) <= 100 else {}
    state, control, t0, toff = pkg.synthetic.config2_inputs(traj, B, seed=4242, **kw)
    res = {}
    for clip in ("0", "1"):
        m = pkg.BatchedTrajectoryTrackingMPC(traj, B, options={"clip_guess": float(clip)})
        ms = []
        for _ in range(4):
            m.reset(); m.set_inputs(state, control, t0, time_offset=toff); m.step_dev(); m.synchronize(); ms.append(m.phase_ms())
        st, it, act, mu = m.solve_info(); pol = m.polish_info(); u = m.get_next_control()
        res[clip] = (u, st, it, pol, np.min(np.array(ms), axis=0))
        m.close()
    a, b = res["0"], res["1"]
    both = (a[3] >= 1) & (b[3] >= 1)
    print(f"{name:14s} clip 0 / 1: interior-point instances {int((a[2] > 0).sum())} / {int((b[2] > 0).sum())}, unsolved {int((~pkg.is_solved(a[1])).sum())} / {int((~pkg.is_solved(b[1])).sum())}, "
          f"rounds mean {a[3][a[3] >= 1].mean():.2f} / {b[3][b[3] >= 1].mean():.2f}, max {a[3].max()} / {b[3].max()}, solve ms {a[4][2]:.3f} / {b[4][2]:.3f}, "
          f"max |du| (both verified) {np.max(np.abs(a[0][both] - b[0][both]) / un):.1e}", flush=True)
