"""Diagnostic: interior-point and polish trace of ONE instance (PG_DEBUG_INSTANCE -> option "diag_instance" of the diagnostic library) at closed-loop step PG_STEP on path PG_PATH -- the closed
loop runs PG_STEP - 1 steps on the device, then the step's phases are called one by one with the diagnostic build of k_solve in place of solve!."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import _load_pkg
pkg = _load_pkg()
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _legacy_env import options_from_env, precision_for
OPTS = options_from_env()
path = os.environ.get("PG_PATH", "EastPaddock"); step = int(os.environ.get("PG_STEP", "27")); inst = int(os.environ["PG_DEBUG_INSTANCE"])
traj = pkg.load_path_fixture(path)
B = 4096
state, control, t0, toff = pkg.synthetic.config2_inputs(traj, B, seed=12345)
if int(os.environ.get("PG_DEC", "0")):                     # PG_DEC=1: the decoupled N = 50 formulation (BASELINE config 5) with the polish and the empty-set rounds switched on
    mpc = pkg.DecoupledTrajectoryTrackingMPC(pkg.X1(), traj, B, N_short=10, N_long=40, walls=bool(int(os.environ.get("PG_WALLS", "0"))), precision="f64-diag", options=OPTS, **(dict(polish=True, cold_guess=8) if int(os.environ.get("PG_DEC_POLISH", "1")) else {}))
else:
    mpc = pkg.BatchedTrajectoryTrackingMPC(traj, B, precision="f64-diag" if os.environ.get("PG_PREC", "f64") == "f64" else os.environ["PG_PREC"], options=OPTS)      # (the trace exists in the fp64 diagnostic library)
other = None
if int(os.environ.get("PG_HJI", "0")):                     # PG_HJI=1 (+ PG_PREC=f32): BASELINE config 3
    mpc.set_hji_cache(*pkg.synthetic.hji_grid_large()); other = pkg.synthetic.other_cars(state, seed=777)
mpc.set_inputs(state, control, t0, other_car_state=other, time_offset=toff)
if step > 1:
    mpc.simulate_(step - 1); mpc.synchronize()
st0, it0, act0, _ = mpc.solve_info(); ps0 = mpc.polish_info()
print(f"before step {step}: instance {inst} status {st0[inst]} iters {it0[inst]} polish {ps0[inst]} active rows {sum(bin(int(m)).count('1') for m in act0[inst])}")
mpc.compute_time_steps_(); mpc.compute_linearization_nodes_(); mpc.update_QP_()
out = np.zeros(B * 9 + 1024, dtype=np.uint64)
rc = mpc.lib.pg_debug_solve_cycles(mpc.h, out.ctypes.data_as(C.c_void_p)); assert rc == 0
tr = out[B * 6:].view(np.float32 if mpc.precision == 'f32' else np.float64).reshape(-1, 4)[:256]
st, it, act, mu = mpc.solve_info(); ps = mpc.polish_info()
print(f"step {step}: instance {inst} status {st[inst]} iters {it[inst]} polish {ps[inst]} active rows {sum(bin(int(m)).count('1') for m in act[inst])}; batch: iters>0 {(it > 0).sum()} max {it.max()}")
w = np.argsort(-it)[:8]; print("slowest instances of the batch (index, iters, polish):", [(int(b), int(it[b]), int(ps[b])) for b in w])
print("interior-point records (mu, alpha_aff, sigma, alpha):")
for k in range(128):
    if tr[k].any(): print(f"  {k:3d}  mu {tr[k, 0]:.3e}  aaff {tr[k, 1]:.3f}  sigma {tr[k, 2]:.3e}  alpha {tr[k, 3]:.3f}")
print("polish checks ((attempt + 2) * 100 + pass [attempt -2 = previous set, -1 = empty set, 0 / 1 = after the interior point], outcome [0 verified, 1 refine, 2 set changed, 3 cycle; +10 = after a refinement], rows in the set, max |t| over them):")
for k in range(128, 256):
    if tr[k].any():
        o = int(tr[k, 1]); r = int(tr[k, 2])
        print(f"  {int(tr[k, 0]):4d}  outcome {o % 100:2d}  rows {r % 1000:3d}  max|t| {tr[k, 3]:.3e}   added {o // 100 % 100} dropped {o // 10000} rows; first added: stage {r // 1000 % 100 - 1} row bit {r // 100000 - 1}")
