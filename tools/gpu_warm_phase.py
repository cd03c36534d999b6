import os, sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from __graft_entry__ import _load_pkg
pkg = _load_pkg(); traj = pkg.load_path_fixture("skidpadoval"); B = 4096
state, control, t0, toff = pkg.synthetic.config2_inputs(traj, B, seed=12345)
mpc = pkg.BatchedTrajectoryTrackingMPC(traj, B)
mpc.set_inputs(state, control, t0, time_offset=toff)
mpc.step_dev(None); mpc.synchronize()
ph = []
for _ in range(10):
    mpc.step_dev(None); mpc.synchronize(); ph.append(mpc.phase_ms())
print("warm step phases (nodes, update_qp, solve):", np.round(np.mean(ph, 0), 3), "iters", mpc.solve_info()[1].mean())
ps = mpc.polish_info(); print("polish rounds of the last warm step:", np.bincount(ps + 1, minlength=6).tolist())
mpc.set_stream(torch.cuda.current_stream().cuda_stream)
for label, reset in (("warm", False), ("cold", True)):
    ts = []
    for _ in range(6):
        if reset: mpc.reset()
        mpc.compute_time_steps_(); mpc.compute_linearization_nodes_(); mpc.update_QP_(); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); mpc.solve_(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    print(label, "k_solve alone:", np.round(ts, 3))
