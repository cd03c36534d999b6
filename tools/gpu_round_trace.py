#!/usr/bin/env python3
"""Which rows join / leave per round in the cold instances of the headline batch that need the most active-set rounds (they set the length of the k_solve launch: a 7-round
instance lives 271 us of a 283 us launch).  For the K slowest instances: the polish checks of the diagnostic build (option "diag_instance" of libpigeon_hip_diag.so) and the
final working set per stage.  usage: tools/gpu_round_trace.py [K] [path]"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import _load_pkg
pkg = _load_pkg()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
path = sys.argv[2] if len(sys.argv) > 2 else "skidpadoval"
traj = pkg.load_path_fixture(path); B = 4096
state, control, t0, toff = pkg.synthetic.config2_inputs(traj, B, seed=12345)
m = pkg.BatchedTrajectoryTrackingMPC(traj, B)
m.step_(state, control, t0, time_offset=toff)
pol = m.polish_info(); st, it, act, mu = m.solve_info()
print("rounds histogram:", np.bincount(pol[pol > 0]))
worst = np.argsort(-pol)[:K]
names = {0: "e_lo", 1: "e_hi", 2: "Fx_lo", 3: "d_hi", 4: "d_lo", 5: "Fx_hi", 6: "env1a", 7: "env1b", 8: "env2a", 9: "env2b", 10: "sig1", 11: "sig2", 12: "dd_hi", 13: "dd_lo", 14: "hji", 15: "sigh"}


def show_set(a):
    out = []
    for s in range(a.shape[0]):
        bits = [names[j] for j in range(16) if (int(a[s]) >> j) & 1 and j not in (10, 11)]
        out.append(f"{s}:{'+'.join(bits)}" if bits else "")
    return " ".join(x for x in out if x)


for b in worst:
    d = pkg.BatchedTrajectoryTrackingMPC(traj, B, precision="f64-diag", options={"diag_instance": int(b)})
    d.set_inputs(state, control, t0, time_offset=toff)
    d.compute_time_steps_(); d.compute_linearization_nodes_(); d.update_QP_()
    out = np.zeros(B * 9 + 1024, dtype=np.uint64)
    rc = d.lib.pg_debug_solve_cycles(d.h, out.ctypes.data_as(C.c_void_p)); assert rc == 0
    tr = out[B * 6:].view(np.float64).reshape(-1, 4)[:256]
    _, _, act_d, _ = d.solve_info(); pol_d = d.polish_info()
    print(f"\ninstance {b}: rounds {pol[b]} (diag kernel {pol_d[b]}); state Ux {state[b, 2]:.2f} Uy {state[b, 3]:.3f} r {state[b, 4]:.3f} dpsi {state[b, 5]:.3f} e {state[b, 1]:.3f}; control delta {control[b, 0]:.4f}")
    print("  final set (sigma pivots left out):", show_set(act[b]))
    for k in range(128, 256):
        if tr[k].any():
            o = int(tr[k, 1]); r = int(tr[k, 2])
            print(f"   pass {int(tr[k, 0]):4d}  outcome {o % 100:2d}  rows held {r % 1000:3d}  max|t| {tr[k, 3]:.2e}  +{o // 100 % 100} -{o // 10000}   first added: stage {r // 1000 % 100 - 1} {names.get(r // 100000 - 1, '-')}")
    del d
