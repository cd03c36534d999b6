"""Diagnostic: cold config-2 solves with the polish first tried from the EMPTY active set (pg_config cold_guess = round cap; 0 = off) -- solve time, share of
instances served without the interior point, agreement of the applied controls with the cold_guess = 0 run."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import _load_pkg
pkg = _load_pkg()
path = os.environ.get("PG_PATH", "skidpadoval")
traj = pkg.load_path_fixture(path)
B = 4096
state, control, t0, toff = pkg.synthetic.config2_inputs(traj, B, seed=12345)
ref = None
prec = os.environ.get("PG_PREC", "f64"); with_hji = bool(int(os.environ.get("PG_HJI", "0")))      # PG_PREC=f32 PG_HJI=1: BASELINE config 3
other = pkg.synthetic.other_cars(state, seed=777) if with_hji else None
for cg in [int(x) for x in os.environ.get("PG_CG", "0,1,2,3,4,6").split(",")]:
    mpc = pkg.BatchedTrajectoryTrackingMPC(traj, B, cold_guess=cg, precision=prec)
    if with_hji:
        mpc.set_hji_cache(*pkg.synthetic.hji_grid_large())
    ts = []
    for rep in range(6):
        mpc.reset()
        mpc.set_inputs(state, control, t0, other_car_state=other, time_offset=toff)
        mpc.step_dev(); mpc.synchronize()
        ts.append(mpc.phase_ms()[2])
    u = mpc.get_next_control()
    st, it, _, _ = mpc.solve_info(); ps = mpc.polish_info()
    if ref is None: ref = u.copy()
    err = np.abs(u - ref) / np.array([0.3, 5000.0, 5000.0])[: u.shape[1]] if u.shape[1] == 3 else np.abs(u - ref)
    print(f"cold_guess={cg}: solve {np.median(ts[1:]):.3f} ms; solved {(st == 1).sum()}, iters==0 {(it == 0).sum()}, iters mean {it.mean():.2f}, polish rounds {np.bincount(ps + 1, minlength=9).tolist()}, "
          f"max |u - u(cg=0)| normalised {err.max():.2e}", flush=True)
    mpc.close()
