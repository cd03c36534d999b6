"""Diagnostic: decoupled N = 50 (BASELINE config 5) -- polish outcome, speed and accuracy against the oracle's exact lateral optimum for a list of settings.
PG_CASES = "tol:polish:rho:ptol[:cold_guess],..."  (tol = where the interior point hands over / stops; cold_guess = rounds of the active-set guess, default 0)"""
import os, sys, time
import numpy as np
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import load_pkg
from oracle import oracle as om
pkg = load_pkg(); sk = pkg.load_path_fixture("skidpadoval")
B = int(os.environ.get("PG_B", "2048")); walls = bool(int(os.environ.get("PG_WALLS", "0")))
state, control, t0, toff = pkg.synthetic.config2_inputs(sk, B, seed=12345)
cases = [c.split(":") for c in os.environ.get("PG_CASES", "1e-12:0:1e6:1e-9,1e-6:1:1e6:1e-9,1e-6:1:1e6:1e-7,1e-6:1:1e5:1e-7,1e-8:1:1e6:1e-7,1e-8:1:1e4:1e-6").split(",")]
ref = None
nthr = min(64, len(os.sched_getaffinity(0)))
for case in cases:
    tol, pol, rho, ptol = case[:4]; cg = int(case[4]) if len(case) > 4 else 0
    tol = float(tol); pol = int(pol)
    kw = dict(polish=True, polish_ipm_tol=tol, polish_rho=float(rho), polish_tol=float(ptol), cold_guess=cg) if pol else dict(polish=False, ipm_tol=tol)
    mpc = pkg.DecoupledTrajectoryTrackingMPC(pkg.X1(), sk, B, N_short=10, N_long=40, walls=walls, **kw)
    u, st, it = mpc.step_(state, control, t0, time_offset=toff)
    ms = []
    for rep in range(3):
        mpc.step_(state, control, t0, time_offset=toff); ms.append(mpc.phase_ms()[2])
    x, _ = mpc.solution(); ps = mpc.polish_info(); qs, us, psn = mpc.nodes(); ts, dt, _ = mpc.time_steps()
    if ref is None and not walls:
        def work(w):
            o = om.OracleDecoupled(N_short=10, N_long=40); o.set_trajectory(sk.data); out = []
            for b in range(w, B, nthr):
                q4 = qs[b][:, 2:6]; p4 = np.stack([qs[b][:, 1], psn[b][:, 1], 0 * psn[b][:, 1], 0 * psn[b][:, 1]], 1)
                sd = o.update_qp(q4, us[b], p4, dt[b]); xe, ye, info = o.solve_exact(sd)
                out.append((b, o.split_x(xe)["delta"], info["status"]))
            return out
        with ThreadPoolExecutor(nthr) as ex:
            res = sum(ex.map(work, range(nthr)), [])
        res.sort(key=lambda r: r[0]); ref = np.stack([r[1] for r in res]); okr = np.array([r[2] for r in res])
        print("oracle solved", int((okr == 1).sum()), "/", B, flush=True)
    line = f"tol {tol:g} polish {pol} rho {rho} ptol {ptol} cold_guess {cg}: solve {min(ms):.3f} ms, iters mean {it.mean():.2f} max {it.max()}, solved {(st == 1).sum()}, polish {np.bincount(ps + 1, minlength=8).tolist()}"
    if ref is not None:
        e2 = np.abs(x[:, 1, 6] - ref[:, 1]); ea = np.max(np.abs(x[:, :, 6] - ref), axis=1); good = st == 1
        line += f"\n   delta_2 error max {e2[good].max():.2e} median {np.median(e2[good]):.2e}; all delta: max {ea[good].max():.2e} p99 {np.percentile(ea[good], 99):.2e}; > 1e-6: {int((e2[good] > 1e-6).sum())}"
    print(line, flush=True)
    mpc.close()
