#!/usr/bin/env python3
"""Timeline of the pipelined nodes + update_QP launch (k_nodes_linearize) of a cold step: per block the entry, the end of its wait for the nodes and its exit on the 100 MHz
wall clock.  Needs the timeline build: make -C pigeon.jl_amd/csrc libpigeon_hip_tl.so; PIGEON_HIP_LIB_DIAG=<that library> python tools/gpu_pipeline_timeline.py [B]"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import _load_pkg
pkg = _load_pkg()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
traj = pkg.load_path_fixture("skidpadoval")
opts = {k: float(v) for k, v in (kv.split("=") for kv in os.environ.get("PG_OPTS", "").split(",") if kv)}      # e.g. PG_OPTS=pipe_pub_short=1
m = pkg.BatchedTrajectoryTrackingMPC(traj, B, precision="f64-diag", options=opts)
state, control, t0, toff = pkg.synthetic.config2_inputs(traj, B, seed=12345)
for _ in range(3):
    m.reset(); m.set_inputs(state, control, t0, time_offset=toff); m.step_dev(); m.synchronize()
nb = B // 32 + (B // 64) * m.N
raw = np.zeros(8 * nb, dtype=np.uint64)
rc = m.lib.pg_debug_pipeline_timeline(m.h, raw.ctypes.data_as(C.c_void_p), C.c_int(nb)); assert rc == 0
tl = raw.reshape(nb, 8); t0_ = tl[:, 0].min()
ent = (tl[:, 0] - t0_) / 100.0; rdy = (np.maximum(tl[:, 1], tl[:, 0]) - t0_) / 100.0; ext = (tl[:, 2] - t0_) / 100.0; kind = tl[:, 3].astype(int)
nodes = kind == 1000
print(f"launch {ext.max():.1f} us; nodes blocks {int(nodes.sum())}: exit mean {ext[nodes].mean():.1f} max {ext[nodes].max():.1f}")
mk = (tl[nodes, 4:8].astype(float) - float(t0_)) / 100.0
print(f"  recurrence, mean over its blocks: trajectory staged {mk[:, 0].mean():.1f} us, measured state seeded {mk[:, 1].mean():.1f}, node 1 done {mk[:, 2].mean():.1f}, node 3 done {mk[:, 3].mean():.1f}")
print("interval: blocks, entry (mean), waited (mean / max), computed (mean), exit (mean / max)")
for t in sorted(set(kind[~nodes])):
    k = kind == t
    print(f"  {t:3d}: {int(k.sum()):4d}  entry {ent[k].mean():7.1f}  waited {np.mean(rdy[k] - ent[k]):6.1f} / {np.max(rdy[k] - ent[k]):6.1f}  computed {np.mean(ext[k] - rdy[k]):6.1f}  exit {ext[k].mean():7.1f} / {ext[k].max():7.1f}")
c = ~nodes
print(f"linearisation wavefronts: waiting {np.sum(rdy[c] - ent[c]) / 1e3:.1f} ms of SIMD time, computing {np.sum(ext[c] - rdy[c]) / 1e3:.1f} ms; nodes {np.sum(ext[nodes] - ent[nodes]) / 1e3:.1f} ms; "
      f"1024 SIMDs x launch = {1024 * ext.max() / 1e3:.1f} ms")
