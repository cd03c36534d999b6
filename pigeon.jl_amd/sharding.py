"""Batch sharding across the GPUs of one node: contiguous block partition, no data-path collective; the only
exchange is the final gather of the B x 3 controls (RCCL all_gather over xGMI when the backend is "nccl")."""
import numpy as np


def shard_range(B, world, rank):
    """Instances [lo, hi) owned by `rank`: GPU g gets [g*B/G, (g+1)*B/G) (remainder spread over the first ranks)."""
    base, rem = divmod(B, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_controls(u_local, world, group=None):
    """all_gather of equally sized [b,3] control shards into [world*b, 3] (torch tensor in, torch tensor out)."""
    import torch
    import torch.distributed as dist
    out = torch.empty((world * u_local.shape[0],) + tuple(u_local.shape[1:]), dtype=u_local.dtype, device=u_local.device)
    dist.all_gather_into_tensor(out, u_local.contiguous(), group=group)
    return out


def gather_controls_ragged(u_local, B, world, rank, group=None):
    """Same for a batch that does not divide evenly: pads every shard to the largest one, gathers, drops the padding."""
    import torch
    sizes = [shard_range(B, world, r)[1] - shard_range(B, world, r)[0] for r in range(world)]
    m = max(sizes)
    pad = torch.zeros((m,) + tuple(u_local.shape[1:]), dtype=u_local.dtype, device=u_local.device)
    pad[: u_local.shape[0]] = u_local
    g = gather_controls(pad, world, group).reshape((world, m) + tuple(u_local.shape[1:]))
    return torch.cat([g[r, : sizes[r]] for r in range(world)], dim=0)
