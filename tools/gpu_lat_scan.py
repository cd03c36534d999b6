"""Diagnostic: solve-phase time and iteration statistics of k_solve_lat at BASELINE configs[4] (N = 50, B = 4096) under an environment knob.
Usage (GPU box): PG_LAT_MU0_COST=30 python tools/gpu_lat_scan.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_pkg
pkg = load_pkg()
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _legacy_env import options_from_env, precision_for      # (the PG_* variables of this tool's usage line become pg_set_option names: the library reads no environment)
OPTS = options_from_env()
traj = pkg.load_path_fixture("skidpadoval")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for walls in (False, True):
    mpc = pkg.DecoupledTrajectoryTrackingMPC(pkg.X1(), traj, B, N_short=10, N_long=40, walls=walls, options=OPTS)
    state, control, t0, toff = pkg.synthetic.config2_inputs(traj, B)
    mpc.step_(state, control, t0, time_offset=toff)
    ms = []
    for _ in range(5):
        mpc.step_(state, control, t0, time_offset=toff); ms.append(mpc.phase_ms())
    st, it, act, mu = mpc.solve_info(); pol = mpc.polish_info()
    ms = np.min(np.array(ms), axis=0)
    print(f"knobs {dict((k, v) for k, v in os.environ.items() if k.startswith('PG_'))} walls {int(walls)}: solve {ms[2]:.3f} ms (step {ms.sum():.3f}), iterations mean {it.mean():.2f} p99 {np.percentile(it, 99):.0f} max {it.max()}, "
          f"per wavefront {it[:B // 4 * 4].reshape(-1, 4).max(1).mean():.2f}, verified {int((pol >= 1).sum())}, status {np.bincount(st, minlength=6).tolist()}", flush=True)
    np.savez(os.path.join(ROOT, 'gpurun_out', f'lat_scan_walls{int(walls)}.npz'), it=it, pol=pol, st=st, mu=mu, **({'edges': mpc.wall_edges()} if walls else {}))
    mpc.close()
