"""Per-phase shader cycles of k_solve's interior point (pg_config.cold_guess = 0: every instance of the config-2 batch goes through it), fp64 diagnostic instantiation.
Runs against whatever tree it sits in (round-4 snapshot under ab/r4tree or the current one): usage  python <tree>/tools/probes/ipm_cycles_probe.py [f64|f32]"""
import ctypes as C, sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_pkg
pkg = load_pkg()
prec = sys.argv[1] if len(sys.argv) > 1 else "f64"
B = 4096
traj = pkg.load_path_fixture("skidpadoval")
mpc = pkg.BatchedTrajectoryTrackingMPC(traj, B, precision="f64-diag" if prec == "f64" else prec, cold_guess=0)      # (round 6: pg_debug_solve_cycles lives in the -DPG_DIAG library only; an fp32 run needs `make EXTRA=-DPG_DIAG libpigeon_hip_f32.so` selected through PIGEON_HIP_LIB_F32)
state, control, t0, toff = pkg.synthetic.config2_inputs(traj, B, seed=12345)
mpc.reset(); mpc.step_(state, control, t0, time_offset=toff)
mpc.reset()
out = np.zeros(B * 9 + 1024, dtype=np.uint64)
rc = mpc.lib.pg_debug_solve_cycles(mpc.h, out.ctypes.data_as(C.c_void_p)); assert rc == 0
out = out[:B * 6].reshape(B, 6)
st, it, act, mu = mpc.solve_info(); pol = mpc.polish_info()
names = ["stage(assemble/step)", "sync", "matrix pass", "vector passes", "forward passes", "prologue"]
tot = out.sum(1).astype(float)
print(ROOT, prec, "iters mean", it.mean(), "polish rounds mean", pol[pol > 0].mean(), "cycles/solve mean", tot.mean())
for i, n in enumerate(names):
    print(f"{n:24s} {out[:, i].mean():12.0f} cycles  {100 * out[:, i].mean() / tot.mean():5.1f} %")
