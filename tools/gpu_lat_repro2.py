"""Diagnostic (round 6): two lateral handles in lockstep; at the first step where their controls differ, print what the differing instances were (status, iterations, polish outcome, previous step's too)."""
import argparse, os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import _load_pkg
pkg = _load_pkg()
ap = argparse.ArgumentParser(); ap.add_argument("--walls", action="store_true"); ap.add_argument("--steps", type=int, default=8); ap.add_argument("--opts", default="{}")
a = ap.parse_args()
opts = json.loads(a.opts)
traj = pkg.load_path_fixture("skidpadoval")
B = 4096
state, control, t0, toff = pkg.synthetic.config2_inputs(traj, B, seed=12345)
ms = [pkg.DecoupledTrajectoryTrackingMPC(pkg.X1(), traj, B, N_short=10, N_long=40, walls=a.walls, options=opts) for _ in range(2)]
for m in ms: m.set_inputs(state, control, t0, time_offset=toff)
prev = None
for k in range(a.steps):
    info = []
    for m in ms:
        s, c, t = m.simulate_(1)[:3]
        st, it, act, mu = m.solve_info(); pol = m.polish_info(); x, sg = m.solution()
        info.append((np.asarray(c).copy(), st.copy(), it.copy(), pol.copy(), x.copy(), np.asarray(s).copy()))
    st0, it0, act0, mu0 = ms[0].solve_info(); st1, it1, act1, mu1 = ms[1].solve_info()
    print(f"   step {k}: iterations differ on {int((it0 != it1).sum())}, status on {int((st0 != st1).sum())}, polish outcome on {int((info[0][3] != info[1][3]).sum())}, active masks on {int((act0 != act1).reshape(B, -1).any(axis=-1).sum())}, mu on {int((mu0 != mu1).sum())}"
          + (f", multipliers on {int((ms[0].multipliers() != ms[1].multipliers()).reshape(B, -1).any(axis=-1).sum())}" if hasattr(ms[0], "multipliers") else ""))
    if hasattr(ms[0], "multipliers"):
        l0, l1 = ms[0].multipliers(), ms[1].multipliers()
        for b in np.where((l0 != l1).reshape(B, -1).any(axis=-1))[0][:4]:
            w = np.argwhere(l0[b] != l1[b])
            print(f"      multipliers of instance {b} differ at (stage, bit) {w[:8].tolist()}: {[(float(l0[b][tuple(i)]), float(l1[b][tuple(i)])) for i in w[:4]]}; status {st0[b]}/{st1[b]} iters {it0[b]}/{it1[b]} polish {info[0][3][b]}/{info[1][3][b]}; active bits there {[int((act0[b][i[0]] >> i[1]) & 1) for i in w[:8]]}")
    dm = np.where(mu0 != mu1)[0]
    for b in dm[:10]: print(f"      mu word of instance {b}: {int(mu0[b]):#x} / {int(mu1[b]):#x}  iters {it0[b]}/{it1[b]}")
    d = np.where((info[0][0] != info[1][0]).any(axis=-1))[0]
    dx = np.where((info[0][4] != info[1][4]).reshape(B, -1).any(axis=-1))[0]
    print(f"step {k}: controls differ on {len(d)}, solutions differ on {len(dx)} instances; status hist {np.bincount(info[0][1], minlength=6).tolist()}, iterations > 0: {int((info[0][2] > 0).sum())}", flush=True)
    if len(dx):
        for b in dx[:12]:
            print(f"   instance {b}: status {info[0][1][b]}/{info[1][1][b]} iters {info[0][2][b]}/{info[1][2][b]} polish {info[0][3][b]}/{info[1][3][b]} max|dx| {np.max(np.abs(info[0][4][b] - info[1][4][b])):.2e}"
                  + (f"   previous step: status {prev[1][b]} iters {prev[2][b]} polish {prev[3][b]}" if prev else ""))
        break
    prev = info[0]
