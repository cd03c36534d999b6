"""k_solve_lat (the lateral formulation's own solve kernel) against the embedding in k_solve (option "lateral_solver" = 2) on the same batches: status, iterations, applied steering,
whole solution, solve-phase time.  Usage (GPU box): python tools/gpu_lat_check.py [B]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_pkg
pkg = load_pkg()
traj = pkg.load_path_fixture("skidpadoval")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096


def run(Ns, Nl, walls, lat, reps=5, polish=None):
    mpc = pkg.DecoupledTrajectoryTrackingMPC(pkg.X1(), traj, B, N_short=Ns, N_long=Nl, walls=walls, polish=polish, options={"lateral_solver": 1.0 if lat else 2.0})
    state, control, t0, toff = pkg.synthetic.config2_inputs(traj, B)
    u, status, iters = mpc.step_(state, control, t0, time_offset=toff)
    ms = []
    for _ in range(reps):
        mpc.step_(state, control, t0, time_offset=toff); ms.append(mpc.phase_ms())
    x, sg = mpc.solution(); st, it, act, mu = mpc.solve_info()
    mpc.close()
    return dict(u=u, status=status, iters=iters, x=x, sg=sg, act=act, mu=mu, ms=np.min(np.array(ms), axis=0))


for Ns, Nl in [(10, 40), (10, 20), (5, 10)]:
    for walls in (False, True):
        a = run(Ns, Nl, walls, True); e = run(Ns, Nl, walls, False); a0 = run(Ns, Nl, walls, True, polish=False)
        both = (a["status"] == 1) & (e["status"] == 1)
        d2 = np.abs(a["x"][:, 1, 6] - e["x"][:, 1, 6])
        dx = np.max(np.abs(a["x"] - e["x"]).reshape(B, -1), axis=1)
        print(f"N={Ns + Nl} walls={int(walls)}: lat status {np.bincount(a['status'], minlength=5)} iters mean {a['iters'].mean():.2f} max {a['iters'].max()} | embedded status {np.bincount(e['status'], minlength=5)} "
              f"iters mean {e['iters'].mean():.2f} max {e['iters'].max()} | |d2 diff| max {d2[both].max():.2e} median {np.median(d2[both]):.1e}, |x diff| max {dx[both].max():.2e} | "
              f"solve ms lat {a['ms'][2]:.3f} (without its polish {a0['ms'][2]:.3f}) embedded {e['ms'][2]:.3f} (phases lat {np.round(a['ms'], 3)})", flush=True)
        bad = np.where(~both)[0]
        if len(bad): print("   not solved by both:", bad[:10], a["status"][bad[:10]], e["status"][bad[:10]], "iters", a["iters"][bad[:10]], "mu", a["mu"][bad[:10]])
