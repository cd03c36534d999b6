"""Diagnostic: per-phase shader cycles inside k_solve_lat (clock64 stamps, accumulated per wavefront; written by lane 0 of every instance).
Usage (GPU box): python tools/gpu_lat_cycles.py [B] [N_long] [walls]"""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import _load_pkg
pkg = _load_pkg()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
Nl = int(sys.argv[2]) if len(sys.argv) > 2 else 40
walls = len(sys.argv) > 3 and sys.argv[3] == "1"
traj = pkg.load_path_fixture("skidpadoval")
mpc = pkg.DecoupledTrajectoryTrackingMPC(pkg.X1(), traj, B, N_short=10, N_long=Nl, walls=walls, precision="f64-diag")      # (pg_debug_solve_cycles lives in the diagnostic library)
state, control, t0, toff = pkg.synthetic.config2_inputs(traj, B)
mpc.set_inputs(state, control, t0, time_offset=toff)
mpc.compute_time_steps_(); mpc.compute_linearization_nodes_(); mpc.update_QP_()      # (no solve yet: the diagnostic launch below is the COLD solve; k_solve_lat warm-starts otherwise)
out = np.zeros(B * 9 + 1024, dtype=np.uint64)
rc = mpc.lib.pg_debug_solve_cycles(mpc.h, out.ctypes.data_as(C.c_void_p)); assert rc == 0
out = out[:B * 6].reshape(B, 6).astype(float)
st, it, act, mu = mpc.solve_info()
wave_it = it.reshape(-1, 4).max(axis=1) if B % 4 == 0 else it          # a wavefront runs until its slowest instance is done
w = out[::4] if B % 4 == 0 else out
names = ["barrier terms (assemble)", "matrix pass", "vector pass", "roll-outs (2 per iteration)", "Newton point / step rules", "stopping rules, prologue, epilogue"]
tot = w.sum(1)
print(f"N = {10 + Nl}, walls = {int(walls)}: iterations mean {it.mean():.2f}, per wavefront (max of 4) {wave_it.mean():.2f}, max {it.max()}; cycles per wavefront mean {tot.mean():.0f}, max {tot.max():.0f}")
for i, n in enumerate(names):
    print(f"{n:36s} {w[:, i].mean():12.0f} cycles  {100 * w[:, i].mean() / tot.mean():5.1f} %   per wavefront-iteration {w[:, i].sum() / wave_it.sum():10.0f}")
print(f"{'total':36s} {tot.mean():12.0f}                  per wavefront-iteration {tot.sum() / wave_it.sum():10.0f}")

pol = mpc.polish_info()
import collections
print("iterations histogram (instances):", sorted(collections.Counter(it.tolist()).items()))
print("polish info histogram:", sorted(collections.Counter(np.asarray(pol).tolist()).items()), " status:", sorted(collections.Counter(np.asarray(st).tolist()).items()))
order = np.argsort(-tot)[:12]
print("slowest wavefronts: cycles, iterations of their 4 instances, polish info")
for wv in order:
    print(f"  {tot[wv]:10.0f}  {it.reshape(-1, 4)[wv].tolist()}  {np.asarray(pol).reshape(-1, 4)[wv].tolist()}")
q = np.percentile(tot, [50, 90, 99, 100])
print("cycles per wavefront: median %.0f  p90 %.0f  p99 %.0f  max %.0f" % tuple(q))
