"""Diagnostic: accuracy of the applied control against the oracle's exact optimum of the same QP data over the WHOLE config-2 batch (4096 instances),
for a list of (interior-point tolerance, polish) settings.  PG_CASES="1e-12:0,1e-12:1,1e-9:1" selects them."""
import os, sys, time
import numpy as np
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import load_pkg, make_oracle
from oracle import oracle as om
pkg = load_pkg(); sk = pkg.load_path_fixture(os.environ.get("PG_PATH", "skidpadoval"))
B = int(os.environ.get("PG_B", "4096"))
s_end = float(sk.s[-1]); state, control, t0, toff = pkg.synthetic.config2_inputs(sk, B, seed=int(os.environ.get("PG_SEED", "12345")), **(dict(s_range=(2.0, 0.4 * s_end)) if s_end <= 90 else {}))
cases = [(float(a), int(b)) for a, b in (c.split(":") for c in os.environ.get("PG_CASES", "1e-12:0,1e-12:1,1e-10:1,1e-8:1,1e-6:1").split(","))]
ref = None
nthr = len(os.sched_getaffinity(0))
for tol, pol in cases:
    # tol: where the interior point hands over to the polish (polish on), or its final tolerance (polish off)
    prec = os.environ.get("PG_PREC", "f64")
    kw = dict(polish=True, polish_ipm_tol=tol) if pol else dict(polish=False, ipm_tol=tol)
    if "PG_RHO" in os.environ: kw["polish_rho"] = float(os.environ["PG_RHO"])
    if "PG_PTOL" in os.environ: kw["polish_tol"] = float(os.environ["PG_PTOL"])
    mpc = pkg.BatchedTrajectoryTrackingMPC(sk, B, precision=prec, **kw)
    u, st, it = mpc.step_(state, control, t0, time_offset=toff)
    ms = []
    for rep in range(3):
        mpc.reset(); mpc.step_(state, control, t0, time_offset=toff); ms.append(mpc.phase_ms()[2])
    qp = mpc.qp_data(); x, _ = mpc.solution(); _, _, act, mu = mpc.solve_info(); ps = mpc.polish_info()
    if ref is None:
        t_ = time.time()
        orcs = [make_oracle(om, sk) for _ in range(nthr)]
        def work(w):
            o = orcs[w]; out = []
            for b in range(w, B, nthr):
                xe, ye, info = o.solve_exact(qp[b]); X = o.split_x(xe)
                out.append((b, X["u"], info["status"], info["polished"]))
            return out
        with ThreadPoolExecutor(nthr) as ex:
            res = sum(ex.map(work, range(nthr)), [])
        res.sort(key=lambda r: r[0])
        ref = np.stack([r[1] for r in res]); ok = np.array([r[2] for r in res])
        print(f"oracle: solved {(ok == 1).sum()}/{B} (polished {sum(1 for r in res if r[3] > 0)}) in {time.time() - t_:.0f} s on {nthr} threads", flush=True)
    e2 = np.max(np.abs(x[:, 1, 6:] - ref[:, 1]), axis=1); ea = np.max(np.abs(x[:, :, 6:] - ref), axis=(1, 2))
    w = int(np.argmax(e2))
    print(f"tol {tol:g} polish {pol}: solve {min(ms):.3f} ms, iters mean {it.mean():.2f} max {it.max()}, solved {(st == 1).sum()}, polish rounds {np.bincount(ps + 1, minlength=6).tolist()} (index 0 = failed, 1 = not run, 2.. = round)\n"
          f"   applied control error: max {e2.max():.2e} (instance {w}, mu {mu[w]:.1e}, iters {it[w]}, polish {ps[w]}) p99.9 {np.percentile(e2, 99.9):.2e} median {np.median(e2):.2e}; "
          f"all controls: max {ea.max():.2e} p99 {np.percentile(ea, 99):.2e};  > 1e-6: {int((e2 > 1e-6).sum())}  > 1e-7: {int((e2 > 1e-7).sum())}  > 1e-8: {int((e2 > 1e-8).sum())}", flush=True)
    mpc.close()
