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


def gather_blocks(local, nbatch, dst=None):
    """Reassemble the batch axis: rank r holds `local` = its contiguous block [hi-lo, ...core]
    (numpy or a torch tensor on its GPU); returns the full [nbatch, ...core] array on every rank
    (dst=None, all_gather) or on rank `dst` only (gather; None elsewhere).  Blocks may differ in
    length by one slice: each rank pads to the longest.  This is the only bulk exchange the path
    has, and only a caller that wants the whole field in one place needs it -- the solves never do."""
    import torch
    import torch.distributed as dist
    is_np = isinstance(local, np.ndarray)
    if not (dist.is_available() and dist.is_initialized()):
        return local.copy() if is_np else local.clone()
    world, rank = dist.get_world_size(), dist.get_rank()
    t = torch.from_numpy(np.ascontiguousarray(local)) if is_np else local.contiguous()
    if dist.get_backend() == 'nccl' and not t.is_cuda:
        t = t.to(torch.device('cuda', torch.cuda.current_device()))
    longest = -(-int(nbatch) // world)
    core = tuple(t.shape[1:])
    pad = torch.zeros((longest,) + core, dtype=t.dtype, device=t.device)
    pad[:t.shape[0]] = t
    if dst is None:
        out = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(out, pad)
    else:
        out = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
        dist.gather(pad, out, dst=dst)
        if rank != dst:
            return None
    full = torch.empty((int(nbatch),) + core, dtype=t.dtype, device=t.device)
    for r in range(world):
        lo, hi = shard_range(nbatch, r, world)
        full[lo:hi] = out[r][:hi - lo]
    return full.cpu().numpy() if is_np else full


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


def sharded_solve_field(solve_local, S, nbatch, dst=None):
    """Like `sharded_solve`, for callers that want the whole solution in one place:
    `solve_local(lo, hi, S_block) -> flags` updates its block S[lo:hi] in place (a view); returns
    (flags [nbatch, 3], S reassembled along the batch axis on every rank, or on `dst` only)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(), dist.get_world_size()
    else:
        rank, world = 0, 1
    lo, hi = shard_range(nbatch, rank, world)
    block = S[lo:hi]
    fl = solve_local(lo, hi, block) if hi > lo else np.zeros((0, 3))
    return gather_flags(fl, nbatch), gather_blocks(block, nbatch, dst=dst)
