"""Experiment (round 6; diagnostic library built with EXTRA=-DLAT_TRIP_MIX): what a trip of a four-instance wavefront of k_solve_lat costs as a function of what its instances
are doing -- cycles per wavefront regressed on its trips, its trips with an instance in the interior point, with one in a polish, with both, and its unfinished instance-trips.
(A deterministic stand-in for "stop the first launch at a fixed TIME": EXPERIMENTS 12.8.)"""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import _load_pkg
pkg = _load_pkg()
B = 4096; walls = len(sys.argv) > 1 and sys.argv[1] == "1"
traj = pkg.load_path_fixture(sys.argv[2] if len(sys.argv) > 2 else "skidpadoval")
mpc = pkg.DecoupledTrajectoryTrackingMPC(pkg.X1(), traj, B, N_short=10, N_long=40, walls=walls, precision="f64-diag")
state, control, t0, toff = pkg.synthetic.config2_inputs(traj, B)
mpc.set_inputs(state, control, t0, time_offset=toff)
mpc.compute_time_steps_(); mpc.compute_linearization_nodes_(); mpc.update_QP_()
out = np.zeros(B * 9 + 1024, dtype=np.uint64)
rc = mpc.lib.pg_debug_solve_cycles(mpc.h, out.ctypes.data_as(C.c_void_p)); assert rc == 0
pc = out[:B * 6].reshape(B, 6).astype(float)[::4]
tl = out[B * 6 + 1024:B * 6 + 1024 + 3 * B].reshape(B, 3)[::4]
m = tl[:, 0]
nI = (m & np.uint64(0xFFFF)).astype(float); nP = ((m >> np.uint64(16)) & np.uint64(0xFFFF)).astype(float); nB = ((m >> np.uint64(32)) & np.uint64(0xFFFF)).astype(float); nA = (m >> np.uint64(48)).astype(float)
trips = ((tl[:, 2] >> np.uint64(32)) & np.uint64(0xFFFF)).astype(float)
cyc = pc.sum(axis=1)
X = np.stack([trips, nI, nP, nB, nA], axis=1)
coef, res, *_ = np.linalg.lstsq(X, cyc, rcond=None)
pred = X @ coef
print(f"walls={int(walls)}: {len(cyc)} wavefronts, trips {trips.mean():.1f} (max {trips.max():.0f}), cycles mean {cyc.mean():.0f}")
print("cycles ~ %.0f trips + %.0f trips-with-IPM + %.0f trips-with-polish + %.0f trips-with-both + %.0f unfinished-instance-trips;  rms residual %.0f (%.1f %% of the mean)" % (*coef, np.sqrt(np.mean((cyc - pred) ** 2)), 100 * np.sqrt(np.mean((cyc - pred) ** 2)) / cyc.mean()))
c1 = np.linalg.lstsq(trips[:, None], cyc, rcond=None)[0]
print("cycles ~ %.0f trips alone: rms residual %.1f %%" % (c1[0], 100 * np.sqrt(np.mean((cyc - trips * c1[0]) ** 2)) / cyc.mean()))
