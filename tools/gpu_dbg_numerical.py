import sys, os; sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from conftest import load_pkg, make_oracle
from oracle import oracle as om
pkg = load_pkg()
traj = pkg.load_path_fixture("vail"); B = 768; seed = 2
rng = np.random.default_rng(seed); s_hi = float(traj.s[-1])
state, control, t0, toff = pkg.synthetic.config2_inputs(traj, B, seed=seed, traj_mode=True, s_range=(3.0, max(8.0, s_hi - 40.0)))
psi = state[:, 2].copy(); e = rng.uniform(-2.5, 2.5, B)
state[:, 0] -= e * np.cos(psi); state[:, 1] -= e * np.sin(psi); state[:, 2] += rng.uniform(-0.6, 0.6, B)
state[:, 3] = np.clip(state[:, 3] * rng.uniform(0.6, 1.8, B), 1.2, 14.5); state[:, 4] = rng.uniform(-1.0, 1.0, B); state[:, 5] += rng.uniform(-0.5, 0.5, B)
X = pkg.X1(); d0 = rng.uniform(-0.95, 0.95, B) * X["delta_max"]; Fx0 = rng.uniform(0.95 * X["Fx_min"], 0.95 * X["Fx_max"], B)
control = np.stack([d0, np.where(Fx0 > 0, 0.0, 0.6) * Fx0, np.where(Fx0 > 0, 1.0, 0.4) * Fx0], axis=1)
toff = np.where(rng.uniform(size=B) < 0.5, 0.0, np.nan)
mpc = pkg.BatchedTrajectoryTrackingMPC(traj, B); orc = make_oracle(om, traj)
u, st, it = mpc.step_(state, control, t0, time_offset=toff)
ok = st == 1
x1, _ = mpc.solution()
state2 = np.stack([orc.plant_step(state[b], control[b], 0.01) for b in range(B)]); control2 = np.where(ok[:, None], u, control); t02 = t0 + 0.01
if not np.all(ok): mpc.reset(mask=~ok)
u2, st2, it2 = mpc.step_(state2, control2, t02, time_offset=toff)
qs, us, ps = mpc.nodes(); qp = mpc.qp_data(); pol = mpc.polish_info(); x2, sg2 = mpc.solution(); _, _, act, mu = mpc.solve_info()
print("step 2 statuses", np.bincount(st2))
for b in np.flatnonzero(st2 == 3):
    print(b, "step1 status", st[b], "iters", it[b], "| step 2: iters", it2[b], "polish", pol[b], "mu", mu[b], "nodes finite", np.isfinite(qs[b]).all() and np.isfinite(us[b]).all(), "qp finite", np.isfinite(qp[b]).all(),
          "max|qp|", np.max(np.abs(qp[b])), "x1 max", np.max(np.abs(x1[b])), "u2", u2[b], "state2", np.round(state2[b], 3))
    ts, dt = orc.time_steps(t02[b])
    oq, ou, op = orc.nodes(state2[b], control2[b], ts, dt, time_offset=toff[b], solved=True, prev_ts=ts, prev_q=x1[b][:, :6], prev_u=x1[b][:, 6:])
    print("   oracle warm nodes vs gpu:", np.max(np.abs(oq - qs[b])), np.max(np.abs(ou - us[b])))
    xe, ye, info = orc.solve_exact(qp[b]); print("   oracle on gpu qp:", info)
