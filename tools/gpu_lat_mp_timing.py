"""Experiment (round 6; diagnostic library built with EXTRA=-DLAT_MP_TIMING): where a stage of k_solve_lat's matrix pass spends its cycles -- clock stamps behind the operand
requests, the 50 multiply-adds of M, G + reciprocal + gain, and the update of P, summed over the launch per instance (the stamps themselves cost ~40 cycles each)."""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import _load_pkg
pkg = _load_pkg()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
walls = len(sys.argv) > 2 and sys.argv[2] == "1"
traj = pkg.load_path_fixture("skidpadoval")
mpc = pkg.DecoupledTrajectoryTrackingMPC(pkg.X1(), traj, B, N_short=10, N_long=40, walls=walls, precision="f64-diag")
state, control, t0, toff = pkg.synthetic.config2_inputs(traj, B)
mpc.set_inputs(state, control, t0, time_offset=toff)
mpc.compute_time_steps_(); mpc.compute_linearization_nodes_(); mpc.update_QP_()
out = np.zeros(B * 9 + 1024, dtype=np.uint64)
rc = mpc.lib.pg_debug_solve_cycles(mpc.h, out.ctypes.data_as(C.c_void_p)); assert rc == 0
pc = out[:B * 6].reshape(B, 6).astype(float)
tl = out[B * 6 + 1024:B * 6 + 1024 + 3 * B].reshape(B, 3)
mp = np.stack([(tl[:, 0] & np.uint64(0xFFFFFFFF)), (tl[:, 0] >> np.uint64(32)), (tl[:, 1] & np.uint64(0xFFFFFFFF)), (tl[:, 1] >> np.uint64(32))], axis=1).astype(float) * 16
trips = ((tl[:, 2] >> np.uint64(32)) & np.uint64(0xFFFF)).astype(float)
w = slice(0, B, 4) if B % 4 == 0 and B > 1024 else slice(0, B)
stages = trips[w] * 50
print(f"B = {B} walls = {int(walls)}: trips per wavefront mean {trips[w].mean():.1f}; matrix pass {pc[w, 1].mean():.0f} cycles per wavefront = {pc[w, 1].sum() / stages.sum():.0f} per stage")
for i, n in enumerate(["operand requests", "M = 1/2 (P + P') X  (50 DPP multiply-adds)", "G, reciprocal, gain, table stores", "P update (25 DPP multiply-adds), copies"]):
    print(f"  {n:48s} {mp[w, i].sum() / stages.sum():8.0f} cycles per stage")
