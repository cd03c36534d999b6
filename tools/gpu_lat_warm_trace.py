"""Diagnostic: the verdicts of the polish inside k_solve_lat during a WARM step (pg_debug_solve_cycles launch with the trace words), for the instances the warm attempt
does not serve.  Usage (GPU box): python tools/gpu_lat_warm_trace.py [B] [warm-up steps] [Nl] [walls]"""
import ctypes as C, os, sys, collections
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_pkg
pkg = load_pkg()
traj = pkg.load_path_fixture("skidpadoval")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
warmup = int(sys.argv[2]) if len(sys.argv) > 2 else 3
Nl = int(sys.argv[3]) if len(sys.argv) > 3 else 40
walls = bool(int(sys.argv[4])) if len(sys.argv) > 4 else False
mpc = pkg.DecoupledTrajectoryTrackingMPC(pkg.X1(), traj, B, N_short=10, N_long=Nl, walls=walls, precision="f64-diag", polish_rho=float(os.environ['PG_RHO']) if 'PG_RHO' in os.environ else None)
state, control, t0, toff = pkg.synthetic.config2_inputs(traj, B)
mpc.set_inputs(state, control, t0, time_offset=toff)
s, c, t, _, _ = mpc.simulate_(warmup)
mpc.set_inputs(s, c, t, time_offset=toff)
mpc.compute_time_steps_(); mpc.compute_linearization_nodes_(); mpc.update_QP_()
out = np.zeros(B * 9 + 1024, dtype=np.uint64)
rc = mpc.lib.pg_debug_solve_cycles(mpc.h, out.ctypes.data_as(C.c_void_p)); assert rc == 0
tl = out[B * 6 + 1024:].reshape(B, 3)
st, it, act, mu = mpc.solve_info(); pol = mpc.polish_info()
def decode(t0_, t1_, n):
    rec = []
    for i in range(min(int(n), 16)):
        nb = (int(t0_) >> (4 * i)) & 15
        nc = (int(t1_) >> (8 * i)) & 255 if i < 8 else -1
        rec.append(("W" if nb & 4 else "c") + ("2" if nb & 8 else "1") + ("C" if nb & 1 else "-") + ("s" if nb & 2 else "u") + (f"{nc}" if nc >= 0 else ""))
    return " ".join(rec)
served = it == 0
print(f"served {served.mean():.4f}; status {np.bincount(st, minlength=6)}")
seqs = collections.Counter()
for b in np.where(~served)[0]:
    # the warm part of the trace only
    recs = decode(tl[b, 0], tl[b, 1], tl[b, 2]).split(" ")
    seqs[" ".join(r for r in recs if r.startswith("W"))] += 1
for k, v in seqs.most_common(40):
    print(f"{v:5d}  {k}")
seqs = collections.Counter()
for b in np.where(served)[0]:
    seqs[decode(tl[b, 0], tl[b, 1], tl[b, 2])] += 1
print("served:")
for k, v in seqs.most_common(15):
    print(f"{v:5d}  {k}")
