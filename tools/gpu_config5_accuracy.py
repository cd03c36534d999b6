"""BASELINE configs[4] (B = 4096 lateral MPCs, N = 50), every instance against the oracle's exact optimum of the same QP data: per-instance table
(error of the applied steering, objective gap, row violation, horizon error, oracle status / verification, iterations, mu, size of the optimum) saved to
gpurun_out/config5_accuracy_<walls>.npz.  Usage (GPU box):  python tools/gpu_config5_accuracy.py [walls] [polish]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_pkg
from test_gpu_decoupled import check_lateral_batch_against_oracle
from oracle import oracle as om

walls = len(sys.argv) > 1 and sys.argv[1] == "1"
polish = None if len(sys.argv) <= 2 else sys.argv[2] == "1"
pkg = load_pkg(); om.build()
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _legacy_env import options_from_env, precision_for      # (the PG_* variables of this tool's usage line become pg_set_option names: the library reads no environment)
OPTS = options_from_env()
traj = pkg.load_path_fixture("skidpadoval")
B, Ns, Nl = 4096, 10, 40
rho = float(os.environ['PG_RHO']) if 'PG_RHO' in os.environ else None          # PG_RHO: polish penalty (default: the library's)
prec = os.environ.get('PG_PREC', 'f64')                                         # PG_PREC=f32: the fp32 library against the oracle's exact optimum of ITS OWN (fp32-rounded) QP data
mpc = pkg.DecoupledTrajectoryTrackingMPC(pkg.X1(), traj, B, N_short=Ns, N_long=Nl, walls=walls, polish=polish, polish_rho=rho, precision=prec, options=OPTS, polish_ipm_tol=float(os.environ['PG_PIT']) if 'PG_PIT' in os.environ else None)
state, control, t0, toff = pkg.synthetic.config2_inputs(traj, B)
if prec == 'f32': state, control = state.astype(np.float32).astype(np.float64), control.astype(np.float32).astype(np.float64)
u, status, iters = mpc.step_(state, control, t0, time_offset=toff)
print(f"precision={prec} walls={walls} rho={rho} far={os.environ.get('PG_LAT_FAR_COST')} pit={os.environ.get('PG_PIT')} phases {np.round(mpc.phase_ms(), 3)}", flush=True)
t = time.time()
res = check_lateral_batch_against_oracle(pkg, om, traj, mpc, B, Ns, Nl, walls, want_more=True)
x, sg = mpc.solution(); st, it, act, mu = mpc.solve_info()
pol = mpc.polish_info()
print(f"oracle sweep {time.time() - t:.1f} s; GPU status {np.bincount(status)}, iterations mean {iters.mean():.2f} max {iters.max()}, polish verified {int((pol >= 1).sum())}/{B} (rounds {np.bincount(pol[pol >= 1])}), failed {int((pol < 0).sum())}")
ok = res[:, 4] == 1
print(f"oracle solved {int(ok.sum())}/{B}, verified KKT point (polished) {int((res[:, 5] >= 1).sum())}/{B}")
for name, sel in [("all oracle-solved", ok), ("oracle verified", ok & (res[:, 5] >= 1)), ("|e*| <= 10 m", ok & (res[:, 6] <= 10.0)), ("|e*| > 10 m", ok & (res[:, 6] > 10.0))]:
    if sel.sum():
        e = res[sel, 0]
        print(f"{name:22s} n={int(sel.sum()):5d}  |d2-d2*| max {e.max():.2e} p99 {np.percentile(e, 99):.2e} median {np.median(e):.1e}  > 1e-6: {int((e > 1e-6).sum())}   objective gap max {res[sel, 1].max():.2e}  horizon max {res[sel, 3].max():.2e}")
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez(os.path.join(ROOT, "gpurun_out", f"config5_accuracy_walls{int(walls)}_polish{polish}.npz"), res=res, status=status, iters=iters, mu=mu, d2=x[:, 1, 6])
