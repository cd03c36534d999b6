"""world_size-2 and world_size-8 gloo tests of the multi-GPU host logic: contiguous batch shards, gather of controls, max-over-ranks time."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _worker(rank, world, port, B, ret):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import load_pkg
    pkg = load_pkg()
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    traj = pkg.load_path_fixture("skidpadoval")
    state, control, t0, toff = pkg.synthetic.config2_inputs(traj, B, seed=5)
    lo, hi = pkg.sharding.shard_range(B, world, rank)
    # stand-in for the per-rank GPU step: any deterministic per-instance map of the inputs (the collective logic is what is under test)
    u_local = torch.from_numpy(np.stack([state[lo:hi, 2], control[lo:hi, 0] + t0[lo:hi], state[lo:hi, 3]], 1))
    if B % world == 0:
        g = pkg.sharding.gather_controls(u_local, world)
    else:
        g = pkg.sharding.gather_controls_ragged(u_local, B, world, rank)
    t = torch.tensor([0.1 * (rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        full = np.stack([state[:, 2], control[:, 0] + t0, state[:, 3]], 1)
        ret["ok"] = bool(np.array_equal(g.numpy(), full)) and abs(float(t) - 0.1 * world) < 1e-12
    dist.barrier(); dist.destroy_process_group()


def _run(B, world=2):
    import socket
    mgr = mp.Manager(); ret = mgr.dict()
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    mp.spawn(_worker, args=(world, port, B, ret), nprocs=world, join=True)
    assert ret.get("ok") is True


def test_gather_even():
    _run(64)


def test_gather_ragged():
    _run(37)


def test_gather_eight_ranks_even_and_ragged():
    """The world size of BASELINE configs[3] (8 ranks, one per GPU of the node): shard ranges, the padded gather of a batch that does not divide by eight, max-over-ranks."""
    _run(4096, world=8)
    _run(1001, world=8)
