"""Diagnostic (round 6): how many trips through k_solve_lat's loop each instance of config 5 needs (interior-point iterations + polish trips: two polish verdicts per trip),
and how many instances / wavefronts of four are still alive after k trips -- the data behind the straggler hand-over.
Usage (GPU box): python tools/gpu_lat_trips.py [B] [N_long] [walls]"""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import _load_pkg
pkg = _load_pkg()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
Nl = int(sys.argv[2]) if len(sys.argv) > 2 else 40
walls = len(sys.argv) > 3 and sys.argv[3] == "1"
traj = pkg.load_path_fixture("skidpadoval")
OPTS = dict(kv.split("=") for kv in os.environ.get("PG_OPTS", "lat_handover=0").split(",") if kv)
mpc = pkg.DecoupledTrajectoryTrackingMPC(pkg.X1(), traj, B, N_short=10, N_long=Nl, walls=walls, precision="f64-diag", options={k: float(v) for k, v in OPTS.items()})      # (pg_debug_solve_cycles lives in the diagnostic library)
state, control, t0, toff = pkg.synthetic.config2_inputs(traj, B)
mpc.set_inputs(state, control, t0, time_offset=toff)
mpc.compute_time_steps_(); mpc.compute_linearization_nodes_(); mpc.update_QP_()
out = np.zeros(B * 9 + 1024, dtype=np.uint64)
rc = mpc.lib.pg_debug_solve_cycles(mpc.h, out.ctypes.data_as(C.c_void_p)); assert rc == 0
cyc = out[:B * 6].reshape(B, 6).astype(float)
tl = out[B * 6 + 1024:].reshape(B, 3)
st, it, act, mu = mpc.solve_info()
nver = (tl[:, 2] & np.uint64(0xFFFFFFFF)).astype(int)
trips = ((tl[:, 2] >> np.uint64(32)) & np.uint64(0xFFFF)).astype(int)          # counted by the kernel (round 6)
trips0 = (tl[:, 2] >> np.uint64(48)).astype(int)                              # trips behind an instance when the resuming launch took it over (0: never handed over)
print(f"N = {10 + Nl}, walls = {int(walls)}: interior-point iterations mean {it.mean():.2f} max {it.max()}; polish verdicts mean {nver.mean():.2f} max {nver.max()}; trips mean {trips.mean():.2f} max {trips.max()}")
wt = trips.reshape(-1, 4).max(axis=1)
print("k : instances alive after k trips, wavefronts (max of 4) alive after k trips")
for k in range(0, int(trips.max()) + 1):
    print(f"{k:3d}: {(trips > k).sum():5d} ({100.0 * (trips > k).mean():5.1f} %)   {(wt > k).sum():5d} ({100.0 * (wt > k).mean():5.1f} %)")
tot = cyc[::4].sum(1)
print("cycles per wavefront-trip (mean over wavefronts of total cycles / max trips of its four):", (tot / np.maximum(wt, 1)).mean())
handed = trips0 > 0
if handed.any():
    names = ["barrier terms (assemble)", "matrix pass", "vector pass", "roll-outs (2 per trip)", "Newton point / step rules", "stopping rules, prologue, epilogue"]
    tr = (trips - trips0)[handed]
    print(f"handed over: {int(handed.sum())} instances at trips {np.bincount(trips0[handed]).nonzero()[0].tolist()} (counts {np.bincount(trips0[handed])[np.bincount(trips0[handed]).nonzero()[0]].tolist()}); "
          f"in the resuming launch (one instance per wavefront): trips mean {tr.mean():.2f} max {tr.max()}, cycles per trip {cyc[handed].sum() / max(tr.sum(), 1):.0f}")
    for i, n in enumerate(names):
        print(f"   {n:36s} per trip {cyc[handed][:, i].sum() / max(tr.sum(), 1):10.0f}")
    first = ~handed
    print(f"first launch: per wavefront-trip {cyc[::4][first[::4]].sum() / np.maximum(trips.reshape(-1, 4).max(axis=1)[first[::4]], 1).sum():.0f} (wavefronts whose first instance was not handed over)")
np.savez(os.path.join("gpurun_out", f"lat_trips_w{int(walls)}.npz"), it=it, nver=nver, trips=trips, trips0=trips0, cyc=cyc, status=st)
