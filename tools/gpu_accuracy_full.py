"""Diagnostic: accuracy of the applied control against the oracle's exact optimum of the same QP data over the WHOLE config-2 batch (4096 instances)."""
import os, sys
import numpy as np
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import load_pkg, make_oracle
from oracle import oracle as om
pkg = load_pkg(); sk = pkg.load_path_fixture("skidpadoval")
B = 4096
state, control, t0, toff = pkg.synthetic.config2_inputs(sk, B, seed=12345)
mpc = pkg.BatchedTrajectoryTrackingMPC(sk, B, ipm_tol=float(os.environ.get("PG_TOL", "1e-12")))
u, st, it = mpc.step_(state, control, t0, time_offset=toff)
ms = []
for rep in range(3):
    mpc.reset(); mpc.step_(state, control, t0, time_offset=toff); ms.append(mpc.phase_ms()[2])
print("solve ms", min(ms), "iters mean", it.mean(), "max", it.max(), "solved", (st == 1).sum())
qp = mpc.qp_data(); x, _ = mpc.solution(); _, _, _, mu = mpc.solve_info()
orcs = [make_oracle(om, sk) for _ in range(16)]
def work(w):
    o = orcs[w]; out = []
    for b in range(w, B, 16):
        xe, ye, info = o.solve_exact(qp[b]); X = o.split_x(xe)
        out.append((b, np.max(np.abs(x[b, 1, 6:] - X["u"][1])), np.max(np.abs(x[b, :, 6:] - X["u"])), info["status"]))
    return out
with ThreadPoolExecutor(16) as ex:
    res = sum(ex.map(work, range(16)), [])
res.sort()
e2 = np.array([r[1] for r in res]); ea = np.array([r[2] for r in res]); ok = np.array([r[3] for r in res])
print(f"oracle solved {(ok==1).sum()}/{B}; applied control error: max {e2.max():.2e} (instance {int(np.argmax(e2))}, mu {mu[int(np.argmax(e2))]:.1e}, iters {it[int(np.argmax(e2))]}) p99.9 {np.percentile(e2, 99.9):.2e} median {np.median(e2):.2e}; all controls: max {ea.max():.2e} p99 {np.percentile(ea, 99):.2e}")
print("instances with applied-control error > 1e-6:", int((e2 > 1e-6).sum()), " > 1e-7:", int((e2 > 1e-7).sum()))
