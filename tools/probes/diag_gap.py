import os, sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from conftest import load_pkg
from test_gpu_decoupled import check_lateral_batch_against_oracle
from oracle import oracle as om
pkg = load_pkg(); om.build()
traj = pkg.load_path_fixture("skidpadoval"); B, Ns, Nl = 4096, 10, 40
for warm in (True, False):
    mpc = pkg.DecoupledTrajectoryTrackingMPC(pkg.X1(), traj, B, N_short=Ns, N_long=Nl, walls=True, warm_polish=warm)
    state, control, t0, toff = pkg.synthetic.config2_inputs(traj, B)
    mpc.set_inputs(state, control, t0, time_offset=toff); s, c, t, _, _ = mpc.simulate_(3)
    mpc.set_inputs(s, c, t, time_offset=toff); mpc.step_dev(); mpc.synchronize()
    st, it, act, mu = mpc.solve_info(); pol = mpc.polish_info()
    res = check_lateral_batch_against_oracle(pkg, om, traj, mpc, B, Ns, Nl, True)
    o = np.argsort(-res[:, 1])[:8]
    print("warm", warm, "worst objective gaps:")
    for b in o: print(f"  b={b} gap {res[b,1]:.2e} d2err {res[b,0]:.2e} rowviol {res[b,2]:.1e} horizon {res[b,3]:.1e} status {st[b]} pol {pol[b]} it {it[b]} |e*|max {res[b,6]:.1f} sigma* {res[b,7]:.1f}")
    print("  gap by class: verified max", res[pol >= 1, 1].max(), "unverified max", res[pol < 1, 1].max() if (pol < 1).any() else None)
