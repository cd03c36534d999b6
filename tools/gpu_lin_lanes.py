"""Diagnostic: update_QP! (launch-per-phase) with the eight tangent directions on 2 / 4 / 8 lanes per interval (PG_LIN_G), for a library built with
-DPG_LIN_WAVES=1 or 2 (two waves per SIMD for k_linearize<K>).  Usage (GPU box): PG_LIN_G=4 python tools/gpu_lin_lanes.py [B]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from __graft_entry__ import _load_pkg
pkg = _load_pkg()
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _legacy_env import options_from_env, precision_for      # (the PG_* variables of this tool's usage line become pg_set_option names: the library reads no environment)
OPTS = options_from_env()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
traj = pkg.load_path_fixture("skidpadoval")
prec = os.environ.get("PREC", "f64")
mpc = pkg.BatchedTrajectoryTrackingMPC(traj, B, precision=precision_for(OPTS, prec), options=OPTS)      # (PG_LIN_G -> "diag_lin_groups": fp64 diagnostic library)
mpc.set_stream(torch.cuda.current_stream().cuda_stream)
state, control, t0, toff = pkg.synthetic.config2_inputs(traj, B, seed=12345, traj_mode=True)
mpc.set_inputs(state, control, t0, time_offset=toff)
mpc.set_pipeline(0)
mpc.compute_time_steps_(); mpc.compute_linearization_nodes_()
for _ in range(3): mpc.update_QP_()
torch.cuda.synchronize()
ts = []
for _ in range(20):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); mpc.update_QP_(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
qp = mpc.qp_data()
h = None
if qp is not None:
    import hashlib; h = hashlib.sha1(np.ascontiguousarray(qp).tobytes()).hexdigest()[:12]
print(f"PG_LIN_G={os.environ.get('PG_LIN_G', '-')} B={B}: update_QP {np.mean(ts):.4f} ms (min {np.min(ts):.4f}) qp sha {h}")
