"""Diagnostic: per-phase shader cycles inside k_solve (diagnostic kernel build with s_memtime stamps; shares, not run time)."""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import _load_pkg
pkg = _load_pkg()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
traj = pkg.load_path_fixture("skidpadoval")
mpc = pkg.BatchedTrajectoryTrackingMPC(traj, B)
state, control, t0, toff = pkg.synthetic.config2_inputs(traj, B, seed=12345)
mpc.step_(state, control, t0, time_offset=toff)
out = np.zeros(B * 6 + 1024, dtype=np.uint64)
rc = mpc.lib.pg_debug_solve_cycles(mpc.h, out.ctypes.data_as(C.c_void_p)); assert rc == 0
out = out[:B * 6].reshape(B, 6)
st, it, act, mu = mpc.solve_info()
names = ["stage(assemble/step)", "sync", "matrix pass", "vector passes", "forward passes", "prologue"]
if os.environ.get("PG_SOLVER") == "quad":
    names = ["stage phases", "matrix pass", "vector pass", "forward passes", "initial roll-out", "-"]
tot = out.sum(1).astype(float)
print("iters mean", it.mean(), "cycles/solve mean", tot.mean())
for i, n in enumerate(names):
    print(f"{n:24s} {out[:, i].mean():12.0f} cycles  {100 * out[:, i].mean() / tot.mean():5.1f} %   per iteration {out[:, i].mean() / it.mean():10.0f}")
