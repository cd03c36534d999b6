"""Diagnostic: per-phase shader cycles inside k_solve (diagnostic kernel build with s_memtime stamps; shares, not run time), then the per-SIMD timeline of a COLD
launch in the product's launch order: wavefront lifetimes by rounds, gaps between consecutive wavefronts of a SIMD, when the SIMDs run out of work, when the long
instances start.  The timeline of the diagnostic kernel is stretched by its stamps (575 against 403 us); for the product's own kernel build
`make -C pigeon.jl_amd/csrc libpigeon_hip_tl.so` and run with PIGEON_HIP_LIB_DIAG=<that library> TL=1 (option "diag_timeline")."""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import _load_pkg
pkg = _load_pkg()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
traj = pkg.load_path_fixture("skidpadoval")
TL = os.environ.get("TL") == "1"
KW = dict(precision="f64-diag", options={"diag_timeline": 1} if TL else {})
mpc = pkg.BatchedTrajectoryTrackingMPC(traj, B, **KW)
state, control, t0, toff = pkg.synthetic.config2_inputs(traj, B, seed=12345)
mpc.step_(state, control, t0, time_offset=toff)
out = np.zeros(B * 9 + 1024, dtype=np.uint64)
rc = mpc.lib.pg_debug_solve_cycles(mpc.h, out.ctypes.data_as(C.c_void_p)); assert rc == 0
out = out[:B * 6].reshape(B, 6)
st, it, act, mu = mpc.solve_info()
names = ["stage(assemble/step)", "sync", "matrix pass", "vector passes", "forward passes", "prologue"]
if os.environ.get("PG_SOLVER") == "quad":
    names = ["stage phases", "matrix pass", "vector pass", "forward passes", "initial roll-out", "-"]
tot = out.sum(1).astype(float)
print("iters mean", it.mean(), "cycles/solve mean", tot.mean())
for i, n in enumerate(names):
    print(f"{n:24s} {out[:, i].mean():12.0f} cycles  {100 * out[:, i].mean() / tot.mean():5.1f} %   per iteration {out[:, i].mean() / it.mean():10.0f}")

# ---- timeline of the launch (k_solve only): per SIMD the wavefronts it ran, their entry / exit on the 100 MHz wall clock, the gaps between them
# (a fresh handle, phases called one by one, the diagnostic solve in place of solve!: a COLD launch, in index order -- the figures above are those of the warm re-solve)
mpc2 = pkg.BatchedTrajectoryTrackingMPC(traj, B, **KW)
mpc2.set_inputs(state, control, t0, time_offset=toff)
mpc2.compute_time_steps_(); mpc2.compute_linearization_nodes_(); mpc2.update_QP_()
raw = np.zeros(B * 9 + 1024, dtype=np.uint64)
rc = mpc2.lib.pg_debug_solve_cycles(mpc2.h, raw.ctypes.data_as(C.c_void_p)); assert rc == 0
mpc = mpc2
cyc = raw[:B * 6].reshape(B, 6).astype(float)
print("cold launch: cycles per instance mean %.0f: " % cyc.sum(1).mean() + ", ".join(f"{n} {cyc[:, i].mean():.0f}" for i, n in enumerate(names)))
tl = raw[B * 6 + 1024:].reshape(B, 3)
if tl[:, 1].any():
    t0_ = tl[:, 0].min(); ent = (tl[:, 0] - t0_).astype(float) / 100.0; ext = (tl[:, 1] - t0_).astype(float) / 100.0          # microseconds
    hw = tl[:, 2] & np.uint64(0xFFFFFFFF); xcc = (tl[:, 2] >> np.uint64(32)) & np.uint64(0xF)
    simd = (hw >> np.uint64(4)) & np.uint64(3); cu = (hw >> np.uint64(8)) & np.uint64(15); sh = (hw >> np.uint64(12)) & np.uint64(1); se = (hw >> np.uint64(13)) & np.uint64(7)
    slot = (((xcc * np.uint64(8) + se) * np.uint64(2) + sh) * np.uint64(16) + cu) * np.uint64(4) + simd
    pol = mpc.polish_info()
    if TL and cyc[:, 5].any():      # -DPG_TIMELINE build, product kernel: wall-clock marks of the first pass through each point (10 ns units since the wavefront's entry)
        mk = cyc / 100.0; life = ext - ent
        for r in (1, 2):
            m = pol == r
            if m.any():
                print(f"instances with {r} round(s) ({int(m.sum())}): prologue done {mk[m, 0].mean():.1f} us, first assembly {mk[m, 1].mean():.1f}, first matrix pass {mk[m, 2].mean():.1f}, "
                      f"first roll-out {mk[m, 3].mean():.1f}, first check {mk[m, 4].mean():.1f}, epilogue starts {mk[m, 5].mean():.1f}, exit {life[m].mean():.1f}")
    print(f"launch: first entry 0, last entry {ent.max():.1f} us, last exit {ext.max():.1f} us; wavefront lifetime mean {np.mean(ext - ent):.1f} us (min {np.min(ext - ent):.1f}, max {np.max(ext - ent):.1f}); distinct SIMDs seen {len(np.unique(slot))}")
    busy = []; gaps = []; nper = []
    for sl_ in np.unique(slot):
        m = np.where(slot == sl_)[0]; o_ = m[np.argsort(ent[m])]
        busy.append(np.sum(ext[o_] - ent[o_])); nper.append(len(o_))
        gaps += list(ent[o_][1:] - ext[o_][:-1])
    busy = np.array(busy); gaps = np.array(gaps)
    print(f"per SIMD: wavefronts {np.mean(nper):.2f} (min {np.min(nper)}, max {np.max(nper)}); busy {busy.mean():.1f} us of {ext.max():.1f} ({100 * busy.mean() / ext.max():.0f} %); gap between consecutive wavefronts on a SIMD: median {np.median(gaps):.2f} us, mean {gaps.mean():.2f}, p99 {np.percentile(gaps, 99):.2f}")
    print("lifetime by rounds:", {int(r): round(float(np.mean((ext - ent)[pol == r])), 1) for r in np.unique(pol)})
    fin = np.sort(np.array([ext[slot == s_].max() for s_ in np.unique(slot)]))
    print(f"time at which a SIMD runs out of work: p10 {np.percentile(fin, 10):.0f}  median {np.median(fin):.0f}  p90 {np.percentile(fin, 90):.0f}  max {fin.max():.0f} us")
    for thr in (4, 6, 8):
        m = pol >= thr
        if m.any(): print(f"instances with >= {thr} rounds: {int(m.sum())}; entry time median {np.median(ent[m]):.0f} us, p90 {np.percentile(ent[m], 90):.0f}, max {ent[m].max():.0f}; exit max {ext[m].max():.0f}")
    last = np.argsort(-ext)[:8]
    print("last wavefronts to finish (exit us, entry us, rounds):", [(round(float(ext[i])), round(float(ent[i])), int(pol[i])) for i in last])
