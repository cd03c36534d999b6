"""Diagnostic: one batch of 4096 on one handle against two half batches on two handles / two streams submitted alternately (the GPU overlaps the latency-bound
nodes kernel of one half with the throughput-bound kernels of the other)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import _load_pkg
pkg = _load_pkg()
import torch
traj = pkg.load_path_fixture("skidpadoval")
B = 4096
state, control, t0, toff = pkg.synthetic.config2_inputs(traj, B, seed=12345)
def run(nh, steps=40):
    hs = []
    for k in range(nh):
        sl = slice(k * B // nh, (k + 1) * B // nh)
        m = pkg.BatchedTrajectoryTrackingMPC(traj, B // nh)
        st = torch.cuda.Stream()
        m.set_stream(st.cuda_stream)
        m.set_inputs(state[sl], control[sl], t0[sl], time_offset=toff[sl])
        hs.append((m, st))
    for _ in range(3):
        for m, st in hs: m.reset(); m.step_dev()
    torch.cuda.synchronize(); a = time.perf_counter()
    for _ in range(steps):
        for m, st in hs: m.reset(); m.step_dev()
    torch.cuda.synchronize(); dt = (time.perf_counter() - a) / steps
    ok = sum(int((m.solve_info()[0] == 1).sum()) for m, _ in hs)
    for m, _ in hs: m.close()
    return dt, ok
for nh in (1, 2, 4):
    dt, ok = run(nh)
    print(f"{nh} handle(s) x {B // nh}: {1e3 * dt:.3f} ms per 4096 ({B / dt / 1e6:.2f} M solves/s), solved {ok}", flush=True)
