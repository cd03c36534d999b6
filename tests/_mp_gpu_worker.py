"""Worker of tests/test_gpu_multiprocess.py: one rank of a world_size-N job (launched by torch.distributed.run).  Every rank steps ITS shard of the
batch with the real HIP path on the GPU(s) present (ranks share cuda:0 on a 1-GPU box), the controls are gathered over gloo, rank 0 saves them."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_pkg  # noqa: E402

out_path, B, precision = sys.argv[1], int(sys.argv[2]), sys.argv[3]
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
pkg = load_pkg()
dev = int(os.environ["LOCAL_RANK"]) % torch.cuda.device_count()
traj = pkg.load_path_fixture("skidpadoval")
state, control, t0, toff = pkg.synthetic.config2_inputs(traj, B, seed=2024)
lo, hi = pkg.sharding.shard_range(B, world, rank)
b = hi - lo
real = np.float64 if precision == "f64" else np.float32
tdt = torch.float64 if precision == "f64" else torch.float32
mpc = pkg.BatchedTrajectoryTrackingMPC(traj, b, device=dev, precision=precision)
d = torch.device("cuda", dev)
ds = torch.from_numpy(state[lo:hi].astype(real)).to(d); dc = torch.from_numpy(control[lo:hi].astype(real)).to(d)
dt0 = torch.from_numpy(t0[lo:hi]).to(d); dto = torch.from_numpy(toff[lo:hi]).to(d)
u = torch.zeros(b, 3, dtype=tdt, device=d)
mpc.set_stream(torch.cuda.current_stream().cuda_stream)
mpc.set_inputs_dev(b, ds.data_ptr(), dc.data_ptr(), dt0.data_ptr(), None, dto.data_ptr())
mpc.step_dev(u.data_ptr())                      # pg_step_dev: the real hot path on this rank's shard
torch.cuda.synchronize()
st, it, _, _ = mpc.solve_info()
g = pkg.sharding.gather_controls_ragged(u.cpu(), B, world, rank) if B % world else pkg.sharding.gather_controls(u.cpu(), world)
ok = torch.tensor([int((st == pkg.SOLVED).all())], dtype=torch.int32)
dist.all_reduce(ok, op=dist.ReduceOp.MIN)
if rank == 0:
    np.savez(out_path, u=g.numpy(), all_solved=int(ok.item()))
dist.barrier(); dist.destroy_process_group()
