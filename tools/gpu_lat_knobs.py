"""Config 5 (B = 4096, N = 50) cold step under the PG_LAT_* knobs of the environment: solve-phase time (min of 5), iteration histogram, verified count, and the distance of the
applied steering from a reference run (default knobs, same process) -- a quick screen before the full oracle sweep of tools/gpu_config5_accuracy.py.
Usage (GPU box): PG_LAT_...=... python tools/gpu_lat_knobs.py [walls 0/1]"""
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
walls = len(sys.argv) > 1 and sys.argv[1] == "1"
Nl = int(sys.argv[2]) if len(sys.argv) > 2 else 40
B = 4096
state, control, t0, toff = pkg.synthetic.config2_inputs(traj, B)
mpc = pkg.DecoupledTrajectoryTrackingMPC(pkg.X1(), traj, B, N_short=10, N_long=Nl, walls=walls, options=OPTS)
ms = []
for _ in range(5):
    mpc.reset(); mpc.set_inputs(state, control, t0, time_offset=toff); mpc.step_dev(); mpc.synchronize(); ms.append(mpc.phase_ms()[2])
st, it, act, mu = mpc.solve_info(); pol = mpc.polish_info(); u = mpc.get_next_control()
ref = os.path.join(ROOT, "gpurun_out", f"lat_knobs_ref_w{int(walls)}_{Nl}.npy")
if os.environ.get("PG_KNOB_REF") == "1": np.save(ref, u[:, 0])
d = np.abs(u[:, 0] - np.load(ref)) if os.path.exists(ref) else np.zeros(B)
knobs = {k: v for k, v in os.environ.items() if k.startswith("PG_LAT")}
print(f"N={10 + Nl} walls={int(walls)} {knobs}: solve {min(ms):.3f} ms | status {np.bincount(st, minlength=6)} | iters mean {it.mean():.2f} p99 {np.percentile(it, 99):.0f} max {it.max()} | >=20: {int((it >= 20).sum())} | "
      f"verified {int((pol >= 1).sum())} rounds {np.bincount(np.clip(pol[pol >= 1], 0, 12))[1:]} | |d2 - ref| max {d.max():.1e}", flush=True)
