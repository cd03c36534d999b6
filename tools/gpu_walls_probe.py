import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import _load_pkg
pkg = _load_pkg()
t = pkg.load_path_fixture("skidpadoval")
wall = float(sys.argv[1]) if len(sys.argv) > 1 else -0.05
tube = pkg.TrajectoryTube(t.t, t.s, t.V, t.A, t.E, t.N, t.psi, t.kappa, edge_L=np.full(len(t), wall), edge_R=np.full(len(t), -4.0))
B = 48; Ns, Nl = 10, 40
rng = np.random.default_rng(5)
s = rng.uniform(5.0, tube.s[-1] - 80.0, B)
E, N, psi, kappa, V, tt = pkg.synthetic.path_pose(tube, s)
e = rng.uniform(-0.5, -0.1, B)
state = np.stack([E - e * np.cos(psi), N - e * np.sin(psi), psi + rng.uniform(-0.05, 0.05, B), V * rng.uniform(0.95, 1.05, B), rng.uniform(-0.1, 0.1, B), kappa * V + rng.uniform(-0.02, 0.02, B)], axis=1)
control = np.stack([rng.uniform(-0.02, 0.02, B), np.zeros(B), rng.uniform(0, 300.0, B)], axis=1)
t0 = tt + rng.uniform(-0.1, 0.1, B); toff = np.zeros(B)
for walls in [False, True]:
    for (ns, nl) in [(10, 20), (10, 40)]:
        mpc = pkg.DecoupledTrajectoryTrackingMPC(pkg.X1(), tube, B, N_short=ns, N_long=nl, walls=walls)
        u, status, iters = mpc.step_(state, control, t0, time_offset=toff)
        st, it, act, mu = mpc.solve_info(); x, sg = mpc.solution()
        print(f"walls={walls} N={ns+nl}: status {np.bincount(status, minlength=5).tolist()} iters {it.tolist()[:16]} mu {np.array2string(mu[:6], precision=2)} max e {x[:, 1:, 5].max():.3f}")
