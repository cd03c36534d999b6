"""Config 3 (B = 4096 coupled + HJI safety row, fp32) cold step under the option "hji_seed" (HJI_SEED=n in the environment of THIS tool; "hji_rounds" likewise): solve-phase time, interior-point iteration histogram, polish rounds of the
instances whose row is violated at the current control, and the distance of the applied controls from a reference run (PG_KNOB_REF=1 stores it).
Usage (GPU box): HJI_SEED=1 python tools/gpu_config3_probe.py [f32|f64]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_pkg
pkg = load_pkg()
prec = sys.argv[1] if len(sys.argv) > 1 else "f32"
traj = pkg.load_path_fixture("skidpadoval"); B = 4096
state, control, t0, toff = pkg.synthetic.config2_inputs(traj, B, seed=12345)
if prec == "f32": state, control = state.astype(np.float32).astype(np.float64), control.astype(np.float32).astype(np.float64)
other = pkg.synthetic.other_cars(state, seed=777)
if prec == "f32": other = other.astype(np.float32).astype(np.float64)
knots, V, g = pkg.synthetic.hji_grid_large()
kw = {} if "PIT" not in os.environ else dict(polish_ipm_tol=float(os.environ["PIT"]))
mpc = pkg.BatchedTrajectoryTrackingMPC(traj, B, precision=prec, **kw, options={"hji_seed": int(os.environ.get("HJI_SEED", "0")), "hji_rounds": int(os.environ.get("HJI_ROUNDS", "0"))})
mpc.set_hji_cache(knots, V, g)
ms = []
for _ in range(5):
    mpc.reset(); mpc.set_inputs(state, control, t0, other_car_state=other, time_offset=toff); mpc.step_dev(); mpc.synchronize(); ms.append(mpc.phase_ms())
ms = np.min(np.array(ms), axis=0)
st, it, act, mu = mpc.solve_info(); pol = mpc.polish_info(); u = mpc.get_next_control()
M, b, Vv = mpc.hji_constraint()
un = np.array([mpc.u_normalization[0], mpc.u_normalization[1], mpc.u_normalization[1]])
hot = (M[:, 0] * control[:, 0] / un[0] + M[:, 1] * (control[:, 1] + control[:, 2]) / un[1] + b) < 0
ref = os.path.join(ROOT, "gpurun_out", f"c3_ref_{prec}.npy")
if os.environ.get("PG_KNOB_REF") == "1": np.save(ref, u)
d = np.max(np.abs(u - np.load(ref)) / un, axis=1) if os.path.exists(ref) else np.zeros(B)
print(f"{prec} seed={os.environ.get('HJI_SEED', '0')} rounds={os.environ.get('HJI_ROUNDS', '0')} pit={os.environ.get('PIT', '-')}: phases {np.round(ms, 3)} | status {np.bincount(st, minlength=6)} | hot {int(hot.sum())}: served by rounds {int((it[hot] == 0).sum())}, rounds hist {np.bincount(np.clip(pol[hot & (it == 0)], 0, 30))[1:]} "
      f"| ipm iters hist {np.bincount(it)} | |u - ref| max {d.max():.1e} (hot {d[hot].max():.1e})", flush=True)
if os.environ.get("PG_C3_DUMP") == "1":            # the verified working sets of the instances with a violated safety row: stage by stage, which rows are held
    names = {0: "Ux>", 1: "Ux<", 2: "Fx>", 3: "d<", 4: "d>", 5: "Fx<", 6: "e0", 7: "e1", 8: "e2", 9: "e3", 10: "s1", 11: "s2", 12: "dd<", 13: "dd>", 14: "HJI", 15: "sH"}
    idx = np.flatnonzero(hot & (pol >= 1) & ((it > 0) if os.environ.get('PG_C3_DUMP_IPM') == '1' else True))
    import collections
    pat = collections.Counter()
    for b_ in idx:
        rows = []
        for k in range(mpc.N):
            m_ = int(act[b_, k]) & ~((1 << 10) | (1 << 11))          # (the sigma >= 0 pivots are held almost everywhere: not shown)
            if m_: rows.append(f"{k}:" + "+".join(names[j] for j in range(16) if (m_ >> j) & 1))
        viol = -(M[b_, 0] * control[b_, 0] / un[0] + M[b_, 1] * (control[b_, 1] + control[b_, 2]) / un[1] + b[b_])
        pat[" ".join(rows)] += 1
        if os.environ.get('PG_C3_DUMP_IPM') == '1' and len(pat) <= 30: print(f"   b={b_} viol {viol:.3f} M {M[b_,0]:.2f} {M[b_,1]:.2f} it {it[b_]} pol {pol[b_]}: {' '.join(rows)[:200]}")
    for k_, v_ in pat.most_common(25): print(f"{v_:4d}  M0>0? -  {k_}")
    sg = np.sign(M[idx, 0]); print("sign(M0) of hot instances:", np.bincount((sg > 0).astype(int)), " M1 sign:", np.bincount((np.sign(M[idx, 1]) > 0).astype(int)))
