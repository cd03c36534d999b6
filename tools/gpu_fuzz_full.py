"""Diagnostic: the wide-random-input stress of tests/test_gpu_fuzz.py with EVERY solved instance checked against the oracle's exact optimum (the test samples 192
of 768) -- cold step and the warm step after it.  PG_FUZZ="path:seed,..."; PG_DUMP="step:instance,..." saves those instances; prints the worst instances (index, error, iterations, polish outcome)."""
import os, sys
import numpy as np
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import _load_pkg
pkg = _load_pkg()
from oracle import oracle as oracle_mod
B = 768
nthr = min(32, len(os.sched_getaffinity(0)))
for spec in os.environ.get("PG_FUZZ", "skidpadoval:1,vail:2,EastPaddock:3").split(","):
    path, seed = spec.split(":"); seed = int(seed)
    traj = pkg.load_path_fixture(path)
    rng = np.random.default_rng(seed)
    s_hi = float(traj.s[-1])
    state, control, t0, toff = pkg.synthetic.config2_inputs(traj, B, seed=seed, traj_mode=True, s_range=(3.0, max(8.0, s_hi - 40.0)))
    psi = state[:, 2].copy()
    e = rng.uniform(-2.5, 2.5, B)
    state[:, 0] -= e * np.cos(psi); state[:, 1] -= e * np.sin(psi)
    state[:, 2] += rng.uniform(-0.6, 0.6, B)
    state[:, 3] = np.clip(state[:, 3] * rng.uniform(0.6, 1.8, B), 1.2, 14.5)
    state[:, 4] = rng.uniform(-1.0, 1.0, B); state[:, 5] += rng.uniform(-0.5, 0.5, B)
    X = pkg.X1()
    d0 = rng.uniform(-0.95, 0.95, B) * X["delta_max"]; Fx0 = rng.uniform(0.95 * X["Fx_min"], 0.95 * X["Fx_max"], B)
    control = np.stack([d0, np.where(Fx0 > 0, 0.0, 0.6) * Fx0, np.where(Fx0 > 0, 1.0, 0.4) * Fx0], axis=1)
    toff = np.where(rng.uniform(size=B) < 0.5, 0.0, np.nan)
    mpc = pkg.BatchedTrajectoryTrackingMPC(traj, B)
    orcs = [oracle_mod.Oracle() for _ in range(nthr)]
    for o in orcs: o.set_trajectory(traj.data)
    for step in range(2):
        u, st, it = mpc.step_(state, control, t0, time_offset=toff)
        qp = mpc.qp_data(); x, _ = mpc.solution(); pol = mpc.polish_info()
        ok = st == pkg.SOLVED
        def work(w):
            out = []
            for b in range(w, B, nthr):
                if ok[b]:
                    xe, ye, info = orcs[w].solve_exact(qp[b])
                    out.append((b, float(np.max(np.abs(x[b, 1, 6:] - orcs[w].split_x(xe)["u"][1]))) if info["status"] == 1 else -1.0, info["iters"]))
            return out
        with ThreadPoolExecutor(nthr) as ex:
            res = sum(ex.map(work, range(nthr)), [])
        errs = np.zeros(B); oit = np.zeros(B, dtype=int)
        for b, e_, i_ in res: errs[b] = e_; oit[b] = i_
        w = np.argsort(-errs)[:4]
        print(f"{path} step {step}: solved {ok.sum()}/{B}, status {np.bincount(st, minlength=5).tolist()}, iters==0 {(it[ok] == 0).sum()}, polish unverified {(pol[ok] < 1).sum()}, > 1e-6: {(errs > 1e-6).sum()}; "
              f"worst (b, err, iters, polish, oracle iters<0 = ADMM fall-back) {[(int(b), float('%.2g' % errs[b]), int(it[b]), int(pol[b]), int(oit[b])) for b in w]}", flush=True)
        if os.environ.get("PG_DUMP"):                       # "step:b,..." -> gpurun_out/fuzz_dump_<path>_<step>_<b>.npz (QP data, solution, solver outcome)
            for sp in os.environ["PG_DUMP"].split(","):
                ds, db = (int(v) for v in sp.split(":"))
                if ds == step:
                    os.makedirs("gpurun_out", exist_ok=True)
                    np.savez(f"gpurun_out/fuzz_dump_{path}_{step}_{db}.npz", qp=qp[db], x=x[db], pol=pol[db], it=it[db], st=st[db], active=mpc.solve_info()[2][db])
        state = np.stack([orcs[0].plant_step(state[b], control[b], 0.01) for b in range(B)]); control = np.where(ok[:, None], u, control); t0 = t0 + 0.01
        if not np.all(ok): mpc.reset(mask=~ok)
    mpc.close()
