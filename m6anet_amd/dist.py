"""Multi-GPU plumbing for the hot path.

Sites are independent, so a job is cut into contiguous, flush-group-aligned site shards (one per
rank / GPU, `shard_plan`), every rank runs encode + pool on its shard with
`engine.set_job_offset(first_site)` so the flush groups and RNG restarts are those of the whole job,
and ONE collective at the end brings site_prob + mod_ratio to rank 0 (`gather_sites`).  There is no
data-path collective.  torch.distributed is only the transport: backend "nccl" is RCCL over xGMI on
MI355X; the same code runs on "gloo" with CPU tensors (tests/test_dist_gloo.py).
"""
import os

import numpy as np

from .engine import shard_plan  # noqa: F401  (re-export)


def init_from_env(backend, device_id=None):
    """Process group from the RANK / WORLD_SIZE / MASTER_* environment torch.distributed.run sets."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if not dist.is_initialized():
        kw = {}
        if device_id is not None:
            kw["device_id"] = device_id
        dist.init_process_group(backend, **kw)
    return dist.get_rank(), dist.get_world_size()


def my_shard(cuts, rank):
    return int(cuts[rank]), int(cuts[rank + 1])


def gather_sites(site, mod, cuts, dst=0, group=None, buffers=None):
    """Gathers every rank's (site_prob float32 [S_r], mod_ratio float64 [S_r]) to rank `dst`.
    Shards are padded to the largest shard so a plain gather works for ragged cuts.  Returns
    (site_all, mod_all) tensors of length cuts[-1] on rank `dst`, (None, None) elsewhere.
    `buffers` = dict reused across calls (avoids reallocating the padded staging tensors)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = np.diff(np.asarray(cuts, np.int64))
    smax = int(sizes.max())
    buffers = buffers if buffers is not None else {}
    key = (smax, site.device, world)
    if buffers.get("key") != key:
        buffers.clear()
        buffers["key"] = key
        buffers["ps"] = torch.zeros(smax, dtype=torch.float32, device=site.device)
        buffers["pm"] = torch.zeros(smax, dtype=torch.float64, device=site.device)
        if rank == dst:
            buffers["gs"] = [torch.empty(smax, dtype=torch.float32, device=site.device) for _ in range(world)]
            buffers["gm"] = [torch.empty(smax, dtype=torch.float64, device=site.device) for _ in range(world)]
    n = site.numel()
    assert n == int(sizes[rank]) and mod.numel() == n
    buffers["ps"][:n].copy_(site)
    buffers["pm"][:n].copy_(mod)
    dist.gather(buffers["ps"], buffers.get("gs") if rank == dst else None, dst=dst, group=group)
    dist.gather(buffers["pm"], buffers.get("gm") if rank == dst else None, dst=dst, group=group)
    if rank != dst:
        return None, None
    site_all = torch.cat([buffers["gs"][r][:int(sizes[r])] for r in range(world)])
    mod_all = torch.cat([buffers["gm"][r][:int(sizes[r])] for r in range(world)])
    return site_all, mod_all
