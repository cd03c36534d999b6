"""Warm start of k_solve_lat in closed loop (decoupled formulation, pg_simulate_dev): per step the phase times of the warm step, how many instances the warm attempt
serves (iters == 0), in which polish round they verify, and the distance of the warm answer from a COLD solve of the same step (a second handle, warm start off,
fed the same inputs).  Usage (GPU box): python tools/gpu_lat_warm.py [B] [steps] [Nl] [walls 0/1]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_pkg
pkg = load_pkg()
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _legacy_env import options_from_env, precision_for      # (the PG_* variables of this tool's usage line become pg_set_option names: the library reads no environment)
OPTS = options_from_env()
traj = pkg.load_path_fixture(os.environ.get("PG_PATH", "skidpadoval"))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
Nl = int(sys.argv[3]) if len(sys.argv) > 3 else 40
walls = bool(int(sys.argv[4])) if len(sys.argv) > 4 else False
Ns = 10

rho = float(os.environ["PG_RHO"]) if "PG_RHO" in os.environ else None
warm = pkg.DecoupledTrajectoryTrackingMPC(pkg.X1(), traj, B, N_short=Ns, N_long=Nl, walls=walls, polish_rho=rho, options=OPTS)
cold = pkg.DecoupledTrajectoryTrackingMPC(pkg.X1(), traj, B, N_short=Ns, N_long=Nl, walls=walls, warm_polish=False)
state, control, t0, toff = pkg.synthetic.config2_inputs(traj, B)
q, u, t = state.copy(), control.copy(), t0.copy()
burn = int(sys.argv[5]) if len(sys.argv) > 5 else 0
if burn:           # closed loop for `burn` steps first (the synthetic starts are far from the path: the first tenths of a second are transients)
    warm.set_inputs(q, u, t, time_offset=toff); q, u, t, _, _ = warm.simulate_(burn)
print(f"burn-in {burn} steps; B={B} N={Ns + Nl} walls={int(walls)} warm_rounds={os.environ.get('PG_LAT_WARM_ROUNDS', 'default')}", flush=True)
act_prev = None
pc8 = np.array([bin(i).count("1") for i in range(65536)], dtype=np.int32)
for k in range(steps):
    # the step the closed loop is about to take, as a timed pg_step_dev on the warm handle (its stored solution is the previous step's) and on the cold one
    warm.set_inputs(q, u, t, time_offset=toff); warm.step_dev(); warm.synchronize(); pw = warm.phase_ms()
    uw = warm.get_next_control(); stw, itw, actw, _ = warm.solve_info(); polw = warm.polish_info()
    cold.reset(); cold.set_inputs(q, u, t, time_offset=toff); cold.step_dev(); cold.synchronize(); pc = cold.phase_ms()
    uc = cold.get_next_control(); stc, itc, actc, _ = cold.solve_info(); polc = cold.polish_info()
    both = (polw >= 1) & (polc >= 1)
    d = np.abs(uw[:, 0] - uc[:, 0])
    served = (itw == 0)
    print(f"step {k:2d}: warm nodes/qp/solve {pw[0]:.3f} {pw[1]:.3f} {pw[2]:.3f} ms | cold solve {pc[2]:.3f} ms | served warm {served.mean():.4f} ({int((~served).sum())} fell back) "
          f"rounds hist {np.bincount(np.clip(polw[served], 0, 9), minlength=6)[:8]} | status {np.bincount(stw, minlength=6)} | |d2 - cold| max over both-verified {d[both].max() if both.any() else float('nan'):.2e} "
          f"(all solved {d[pkg.is_solved(stw) & pkg.is_solved(stc)].max():.2e}) | same set {np.mean(np.all(actw == actc, axis=1)):.3f}", flush=True)
    if act_prev is not None:       # how far is the previous step's final working set from this step's (cold, verified) one -- as it stands, and with the short horizon shifted by one stage
        shift = act_prev.copy(); shift[:, :Ns - 1] = act_prev[:, 1:Ns]
        d0 = pc8[act_prev ^ actc].sum(axis=1); d1 = pc8[shift ^ actc].sum(axis=1)
        dl0 = pc8[(act_prev ^ actc)[:, Ns:]].sum(axis=1)
        for name, sel in (("served", served & both), ("fell back", ~served & both)):
            if sel.any():
                print(f"      {name:9s} n={int(sel.sum()):4d}: rows differing prev->new mean {d0[sel].mean():.2f} p50 {np.percentile(d0[sel], 50):.0f} p90 {np.percentile(d0[sel], 90):.0f} max {d0[sel].max()} | with shift mean {d1[sel].mean():.2f} "
                      f"p90 {np.percentile(d1[sel], 90):.0f} max {d1[sel].max()} | long part only mean {dl0[sel].mean():.2f} | prev status5 share {np.mean(st_prev[sel] == 5):.2f}", flush=True)
    act_prev = actw.copy(); st_prev = stw.copy()
    # advance the closed loop by one step on the warm handle's device plant: re-install the inputs (the timed step above already consumed them) and simulate ONE step
    # -- which repeats the solve just done (warm from itself: trivially served) and advances the plant
    s2, c2, t2, _, _ = warm.simulate_(1)
    q, u, t = s2, c2, t2
