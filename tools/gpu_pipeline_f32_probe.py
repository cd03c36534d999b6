"""Diagnostic: where do the fp32 results of the pipelined nodes + update_QP launch differ from the launch-per-phase sequence?  (fp64: nowhere.)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge._load_pkg()
traj = pkg.load_path_fixture("skidpadoval")
n = 2048
state, control, t0, toff = pkg.synthetic.config2_inputs(traj, n, seed=31)
for prec in ("f64", "f32"):
    out = {}
    for piped in (False, True):
        mpc = pkg.BatchedTrajectoryTrackingMPC(traj, n, precision=prec)
        mpc.set_pipeline(piped)
        mpc.set_inputs(state, control, t0, time_offset=toff)
        mpc.step_dev(); mpc.synchronize()
        qs, us, ps = mpc.nodes()
        out[piped] = dict(qs=qs.copy(), us=us.copy(), ps=ps.copy(), qp=mpc.qp_data().copy(), u=mpc.get_next_control().copy())
        mpc.close()
    for k in out[True]:
        a, b = out[False][k], out[True][k]
        d = np.abs(a - b)
        print(prec, k, "identical" if np.array_equal(a, b, equal_nan=True) else f"max diff {np.nanmax(d):.3e} at {np.unravel_index(np.nanargmax(d), d.shape)}; differing entries {int((a != b).sum())} of {a.size}")
    if prec == "f32":
        a, b = out[False]["qp"], out[True]["qp"]
        cols = np.where((a != b).any(axis=0))[0]
        print("qp columns that differ:", cols[:60], "... count", len(cols))
        a, b = out[False]["qs"], out[True]["qs"]
        print("qs differing (node, comp):", sorted(set(zip(*np.where((a != b))[1:])))[:40])
        a, b = out[False]["us"], out[True]["us"]
        print("us differing (node, comp):", sorted(set(zip(*np.where((a != b))[1:])))[:40])
