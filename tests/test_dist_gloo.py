"""Batch-axis sharding across ranks, world_size 2, gloo on CPU (the N > 1 path of bench.py)."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from xinvert_amd import dist as xdist


def test_shard_range_partitions():
    for nb in (1, 5, 8, 64, 120):
        for world in (1, 2, 3, 8):
            spans = [xdist.shard_range(nb, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == nb
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert xdist.shard_range(64, 3, 8) == (24, 32) and xdist.shard_range(120, 7, 8) == (105, 120)


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, nbatch, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import torch.distributed as dist
    xdist.init_process_group('gloo')

    def solve_local(lo, hi):
        # stand-in for the per-rank GPU solve: flags that encode the global slice index
        return np.stack([[0.0, 1e-9 * (m + 1), 100.0 + m] for m in range(lo, hi)])

    allf = xdist.sharded_solve(solve_local, nbatch)
    q.put((rank, allf))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('nbatch', [7, 8])
def test_sharded_solve_gathers_flags_world2(nbatch):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, nbatch, q)) for r in range(2)]
    [p.start() for p in procs]
    res = dict(q.get(timeout=120) for _ in range(2))
    [p.join(60) for p in procs]
    want = np.stack([[0.0, 1e-9 * (m + 1), 100.0 + m] for m in range(nbatch)])
    for r in range(2):
        assert np.array_equal(res[r], want)


def _worker_field(rank, world, port, nbatch, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import torch.distributed as dist
    xdist.init_process_group('gloo')
    S = np.zeros((nbatch, 5, 7))                        # every rank starts from the same initial guess

    def solve_local(lo, hi, block):
        # stand-in for the per-rank GPU solve: writes its block IN PLACE, slice m gets a pattern of m
        for k, m in enumerate(range(lo, hi)):
            block[k] = (m + 1) * np.arange(35.0).reshape(5, 7)
        return np.stack([[0.0, 1e-9 * (m + 1), 100.0 + m] for m in range(lo, hi)])

    allf, full = xdist.sharded_solve_field(solve_local, S, nbatch)
    _, on0 = xdist.sharded_solve_field(solve_local, S, nbatch, dst=0)
    q.put((rank, allf, full, on0))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('nbatch', [7, 8])
def test_sharded_solve_reassembles_S_world2(nbatch):
    """The solution blocks -- not only the flags -- come back together along the batch axis, in the
    order of the reference's slice loop (core.py:129), on every rank or on one."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_field, args=(r, 2, port, nbatch, q)) for r in range(2)]
    [p.start() for p in procs]
    res = {}
    for _ in range(2):
        r, allf, full, on0 = q.get(timeout=120)
        res[r] = (allf, full, on0)
    [p.join(60) for p in procs]
    want = np.stack([(m + 1) * np.arange(35.0).reshape(5, 7) for m in range(nbatch)])
    wantf = np.stack([[0.0, 1e-9 * (m + 1), 100.0 + m] for m in range(nbatch)])
    for r in range(2):
        assert np.array_equal(res[r][0], wantf) and np.array_equal(res[r][1], want)
    assert np.array_equal(res[0][2], want) and res[1][2] is None


def test_gather_flags_single_process():
    f = np.arange(12.0).reshape(4, 3)
    assert np.array_equal(xdist.gather_flags(f, 4), f)
