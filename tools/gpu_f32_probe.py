"""Diagnostic: fp32 library against the fp64 library on the same (float-representable) inputs."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import _load_pkg
pkg = _load_pkg()
traj = pkg.load_path_fixture("skidpadoval")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
state, control, t0, toff = pkg.synthetic.config2_inputs(traj, B, seed=12345)
state = state.astype(np.float32).astype(np.float64); control = control.astype(np.float32).astype(np.float64)
m64 = pkg.BatchedTrajectoryTrackingMPC(traj, B)
u64, st64, it64 = m64.step_(state, control, t0, time_offset=toff)
q64, uu64, p64 = m64.nodes(); qp64 = m64.qp_data(); x64, _ = m64.solution()
un = np.array([m64.u_normalization[0], m64.u_normalization[1], m64.u_normalization[1]])
for tol in [float(a) for a in sys.argv[2:]] or [1e-4, 1e-5, 1e-6]:
    m32 = pkg.BatchedTrajectoryTrackingMPC(traj, B, precision="f32", ipm_tol=tol)
    u32, st32, it32 = m32.step_(state, control, t0, time_offset=toff)
    q32, uu32, p32 = m32.nodes(); qp32 = m32.qp_data(); x32, _ = m32.solution()
    ts64 = m64.time_steps()[0]; ts32 = m32.time_steps()[0]
    print(f"tol {tol:g}: solved {(st32 == 1).sum()}/{B} status hist {np.bincount(st32, minlength=5).tolist()} iters mean {it32.mean():.2f} max {it32.max()}"
          f" | ts equal {np.array_equal(ts64, ts32)} nodes err {np.max(np.abs(q32 - q64)):.2e} qp rel err {np.max(np.abs(qp32 - qp64)) / np.max(np.abs(qp64)):.2e}"
          f" | u err (normalised) max {np.max(np.abs(u32 - u64) / un):.2e} median {np.median(np.max(np.abs(u32 - u64) / un, axis=1)):.2e}"
          f" x[1] err {np.max(np.abs(x32[:, 1] - x64[:, 1])):.2e}  solve ms {m32.phase_ms() if False else ''}", flush=True)
    m32.close()
m32 = pkg.BatchedTrajectoryTrackingMPC(traj, B, precision="f32")
u32, st32, it32 = m32.step_(state, control, t0, time_offset=toff)
q32, uu32, p32 = m32.nodes(); sep32 = m32.path_coordinates(); sep64 = m64.path_coordinates()
print("sep err (s, e, t):", np.max(np.abs(sep32 - sep64), axis=0))
print("node q err per component:", np.max(np.abs(q32 - q64), axis=(0, 1)))
print("node q err at node 0:", np.max(np.abs(q32[:, 0] - q64[:, 0]), axis=0))
print("node u err:", np.max(np.abs(uu32 - uu64), axis=(0, 1)), " p err:", np.max(np.abs(p32 - p64), axis=(0, 1)))
qp32 = m32.qp_data(); G32 = orc = None
from oracle import oracle as om
o = om.Oracle(); o.set_trajectory(traj.data)
U32 = o.unpack_sd(qp32[0]); U64 = o.unpack_sd(qp64[0])
for k in U64:
    a = np.array([np.max(np.abs(np.asarray(o.unpack_sd(qp32[b])[k]) - np.asarray(o.unpack_sd(qp64[b])[k]))) for b in range(0, B, 16)])
    print(f"qp[{k}] max abs err {a.max():.2e}  (scale {np.max(np.abs(U64[k])):.2e})")
print("phase ms f32:", m32.phase_ms(), " f64:", m64.phase_ms())
x32, _ = m32.solution(); st, it, act, mu = m32.solve_info()
worst = []
for b in range(0, B, 64):
    xe, ye, info = o.solve_exact(qp32[b])
    worst.append(np.max(np.abs(x32[b, 1, 6:] - o.split_x(xe)["u"][1])))
print("f32 solver vs exact optimum of ITS OWN qp data (first control, normalised): max", np.max(worst), "median", np.median(worst))
for prec, m in [("f32", m32), ("f64", m64)]:
    ph = []
    for rep in range(5):
        m.reset(); m.step_(state, control, t0, time_offset=toff); ph.append(m.phase_ms())
    ph = np.min(np.array(ph), axis=0)
    print(prec, "phase ms (min of 5):", ph, "total", ph.sum(), "->", B / ph.sum() / 1e3, "M solves/s")
