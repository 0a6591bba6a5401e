"""Batch-axis sharding across GPUs: one process per GPU, torch.distributed (RCCL on ROCm).

The hot path shards only along the outer (time / level / member) axis -- slices are fully
independent (reference core.py:129: the loop body carries no cross-slice state).  Each rank
solves a contiguous block of slices on its own GPU with no data-path collective; the single
exchange step is a gather of the per-slice convergence flags (3 doubles per slice), done with
one all_gather over RCCL/xGMI (backend 'nccl') or gloo on CPU for tests.
"""
import os

import numpy as np


def shard_range(nbatch, rank, world):
    """Contiguous block partition of `nbatch` slices: ranks < nbatch % world get one extra."""
    q, r = divmod(int(nbatch), int(world))
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def env_rank_world():
    return (int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0')),
            int(os.environ.get('WORLD_SIZE', '1')))


def init_process_group(backend=None):
    """Join the job described by RANK / WORLD_SIZE / MASTER_* (torchrun contract)."""
    import torch
    import torch.distributed as dist
    rank, local, world = env_rank_world()
    # XINV_DIST_FORCE_INIT=1: join even as a single rank (exercises the RCCL path on a 1-GPU box)
    force = os.environ.get('XINV_DIST_FORCE_INIT') == '1'
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            backend = os.environ.get('XINV_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        if backend == 'nccl':
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def gather_flags(local_flags, nbatch, device=None):
    """All-gather the per-slice flags: local [nb_local, 3] -> global [nbatch, 3] on every rank.

    Shards may differ in length by one slice, so each rank pads to the longest shard; payload is
    <= 24 B per slice -- latency-bound, xGMI link bandwidth is irrelevant."""
    import torch
    import torch.distributed as dist
    local_flags = np.asarray(local_flags, dtype=np.float64).reshape(-1, 3)
    if not (dist.is_available() and dist.is_initialized()):
        return local_flags.copy()
    world, rank = dist.get_world_size(), dist.get_rank()
    longest = -(-int(nbatch) // world)
    pad = np.zeros((longest, 3))
    pad[:local_flags.shape[0]] = local_flags
    if device is None:
        device = torch.device('cuda', torch.cuda.current_device()) \
            if dist.get_backend() == 'nccl' else torch.device('cpu')
    t = torch.from_numpy(pad).to(device)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    res = np.zeros((int(nbatch), 3))
    for r in range(world):
        lo, hi = shard_range(nbatch, r, world)
        res[lo:hi] = out[r].cpu().numpy()[:hi - lo]
    return res


def sharded_solve(solve_local, nbatch):
    """Run `solve_local(lo, hi) -> flags[hi-lo, 3]` on this rank's block, gather all flags."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(), dist.get_world_size()
    else:
        rank, world = 0, 1
    lo, hi = shard_range(nbatch, rank, world)
    fl = solve_local(lo, hi) if hi > lo else np.zeros((0, 3))
    return gather_flags(fl, nbatch)
