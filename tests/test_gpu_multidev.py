"""In-process multi-device batched entry and the chunked upload / solve / download pipeline of the
host-pointer entry points (include/xinv.h: xinv_options.ndev / device_ids / host_chunk).

The GPU box has one GPU, so the device list names device 0 twice: two host threads, two contiguous
blocks of the batch axis, the same code path as two distinct GPUs (the threads then serialise on
the per-device lock).  What is checked is the part that can go wrong: block boundaries, pointer
offsets with shared (stride-0) and per-member arrays, S and flags landing in the caller's arrays.
"""
import numpy as np
import pytest

import util
from util import rand2d, rand3d, run_oracle

pytestmark = pytest.mark.gpu
C2 = 2


def _members(nb, kind='gen2d', yc=40, xc=130):
    if kind == 'std3d':
        return [rand3d(9, 20, 66, 'fixed', 'periodic', 1, seed=s) for s in range(nb)]
    return [rand2d(kind, yc, xc, 'fixed', 'periodic', 0, 1, seed=s) for s in range(nb)]


@pytest.mark.parametrize('nb,devs', [(5, [0, 0]), (7, [0, 0, 0]), (2, [0, 0, 0, 0]), (1, [0, 0])])
def test_device_list_splits_batch_in_contiguous_blocks(nb, devs):
    ps = _members(nb)
    for q in ps:                                  # A shared by every member, the rest per member
        q['coefs'][0] = ps[0]['coefs'][0]
    S1, f1, s1 = util.run_hip_batched(ps, 60, 1e-6, shared=(0,))
    S2, f2, s2 = util.run_hip_batched(ps, 60, 1e-6, shared=(0,), devices=devs)
    assert s1['devices'] == 1 and s2['devices'] == min(len(devs), nb)
    assert np.array_equal(S1, S2) and np.array_equal(f1, f2)
    for m in (0, nb - 1):
        So, flo = run_oracle(ps[m], 60, 1e-6, C2)
        assert np.array_equal(S2[m], So) and f2[m][2] == flo[2]
    assert len(set(f2[:, 2])) > 1 or nb == 1      # members stop at different sweeps: flags are per slice


def test_device_list_3d_and_all_devices():
    ps = _members(4, 'std3d')
    S1, f1, _ = util.run_hip_batched(ps, 20, 0.0)
    S2, f2, s2 = util.run_hip_batched(ps, 20, 0.0, devices=[0, 0])
    S3, f3, s3 = util.run_hip_batched(ps, 20, 0.0, devices='all')
    assert np.array_equal(S1, S2) and np.array_equal(S1, S3) and np.array_equal(f1, f2) and np.array_equal(f1, f3)
    assert s2['devices'] == 2 and s3['devices'] >= 1


def test_bad_device_id_is_an_argument_error():
    from xinvert_amd import _lib
    ps = _members(2)
    with pytest.raises(_lib.XinvError, match='no such device'):
        util.run_hip_batched(ps, 5, 0.0, devices=[0, 99])


@pytest.mark.parametrize('chunk', [1, 2, 3, 0])
def test_host_chunk_pipeline_is_result_neutral(chunk):
    """Upload / sweeps / download overlap over member chunks: any chunking gives the same bits."""
    ps = _members(7)
    Sref, fref, sref = util.run_hip_batched(ps, 40, 1e-5, host_chunk=7)
    S, fl, st = util.run_hip_batched(ps, 40, 1e-5, host_chunk=chunk)
    assert sref['host_chunks'] == 1
    if chunk:
        assert st['host_chunks'] == -(-7 // chunk)
    assert np.array_equal(S, Sref) and np.array_equal(fl, fref)
    assert st['wall_ms'] > 0


def test_host_chunks_large_batch_strided_members():
    """Members that are not contiguous on the host (batch stride > slice size) through the chunked
    path, row-constant coefficients expanded once for all chunks."""
    import ctypes
    from xinvert_amd import _lib
    L = _lib.require_gpu()
    nb, yc, xc = 6, 30, 64
    ps = [rand2d('std2d', yc, xc, 'extend', 'periodic', 0, 1, seed=10 + s) for s in range(nb)]
    n = yc * xc
    pad = n + 24
    Sbuf = np.zeros(nb * pad); Fbuf = np.zeros(nb * pad)
    for m, q in enumerate(ps):
        Sbuf[m * pad:m * pad + n] = q['S0'].ravel()
        Fbuf[m * pad:m * pad + n] = q['coefs'][3].ravel()
    Arow = np.ascontiguousarray(np.linspace(0.8, 1.2, yc))          # one value per row
    C = np.ascontiguousarray(ps[0]['coefs'][2])
    fl = np.tile(np.array([0., 1., 0.]), (nb, 1))
    o = _lib.options(rowconst_mask=1, host_chunk=2)
    p = ps[0]
    rc = L.xinv_standard_2d_f64_batched(
        _lib.hptr(Sbuf), _lib.hptr(Arow), None, _lib.hptr(C), _lib.hptr(Fbuf), nb,
        _lib.strides_arg([pad, 0, 0, 0, pad]), yc, xc, p['dely'], p['delx'], _lib.bc('extend'),
        _lib.bc('periodic'), p['delxSqr'], p['ratioQtr'], p['ratioSqr'], p['optArg'], p['undef'],
        _lib.hptr(fl), 25, 0.0, ctypes.byref(o))
    _lib.check(rc)
    assert _lib.last_stats()['host_chunks'] == 3
    A = np.repeat(Arow[:, None], xc, axis=1)
    for m, q in enumerate(ps):
        r = dict(q); r['coefs'] = [A, np.zeros((yc, xc)), C, q['coefs'][3]]
        So, flo = run_oracle(r, 25, 0.0, C2)
        assert np.array_equal(Sbuf[m * pad:m * pad + n].reshape(yc, xc), So)
        assert (Sbuf[m * pad + n:(m + 1) * pad] == 0).all()          # the gaps are never touched


# ------------------------------------------------------------------ distinct GPUs, when the box has them
def test_distinct_devices_split_and_report():
    """With more than one GPU visible the device list names DIFFERENT GPUs: same bits as one device, the
    statistics say how many were used, every device's block lands in the caller's arrays."""
    from xinvert_amd import _lib
    ndev = _lib.load().xinv_device_count()
    if ndev < 2:
        pytest.skip('one GPU visible: distinct device ids need a multi-GPU node (device 0 named twice is covered above)')
    nb = 2 * ndev + 1
    ps = _members(nb)
    S1, f1, s1 = util.run_hip_batched(ps, 60, 1e-6)
    S2, f2, s2 = util.run_hip_batched(ps, 60, 1e-6, devices=list(range(ndev)))
    S3, f3, s3 = util.run_hip_batched(ps, 60, 1e-6, devices='all')
    assert s1['devices'] == 1 and s2['devices'] == ndev and s3['devices'] == ndev
    assert np.array_equal(S1, S2) and np.array_equal(f1, f2) and np.array_equal(S1, S3) and np.array_equal(f1, f3)
    assert s2['wall_ms'] > 0 and s2['h2d_ms'] > 0 and s2['sweep_ms'] >= 0


# ------------------------------------------------------------------ one process per GPU: RCCL and bench.py
def _run(cmd, env_extra, timeout=900):
    import os
    import subprocess
    env = dict(os.environ); env.update(env_extra)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env,
                          cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def test_gather_flags_over_rccl_single_rank():
    """The one collective of the path (all_gather of the per-slice flags) on the backend production uses:
    a 1-rank `nccl` (= RCCL) group on this GPU, in a subprocess (xinvert_amd.dist, XINV_DIST_FORCE_INIT)."""
    import sys
    code = (
        "import numpy as np, torch, torch.distributed as dist\n"
        "from xinvert_amd import dist as xdist\n"
        "rank, local, world = xdist.init_process_group()\n"
        "assert dist.is_initialized() and dist.get_backend() == 'nccl' and world == 1\n"
        "f = np.arange(21.0).reshape(7, 3)\n"
        "g = xdist.gather_flags(f, 7)\n"
        "assert np.array_equal(g, f)\n"
        "blk = torch.arange(70.0, dtype=torch.float64, device='cuda').reshape(7, 2, 5)\n"
        "full = xdist.gather_blocks(blk, 7)\n"
        "assert full.is_cuda and torch.equal(full, blk)\n"
        "t = torch.ones(1, device='cuda'); dist.all_reduce(t); assert float(t.item()) == 1.0\n"
        "dist.destroy_process_group(); print('RCCL_OK')\n")
    import socket
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    out = _run([sys.executable, '-c', code], dict(XINV_DIST_FORCE_INIT='1', RANK='0', LOCAL_RANK='0', WORLD_SIZE='1',
                                                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port)))
    assert out.returncode == 0 and 'RCCL_OK' in out.stdout, (out.stdout[-2000:], out.stderr[-3000:])


def test_bench_self_launches_two_ranks():
    """`bench.py --gpus 2` started WITHOUT a torchrun environment starts its own two ranks; here they share the
    one GPU (XINV_FORCE_DEVICE=0) and exchange the flags over gloo.  The JSON line reports the ranks that
    took part, the real C4 batch split in two blocks, every member swept to the end."""
    import json
    import os
    import sys
    env = {k: None for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    e = dict(XINV_FORCE_DEVICE='0', XINV_DIST_BACKEND='gloo')
    full = dict(os.environ); full.update(e)
    for k in env:
        full.pop(k, None)
    import subprocess
    out = subprocess.run([sys.executable, 'bench.py', '--config', 'c4', '--members', '4', '--gpus', '2', '--steps', '2',
                          '--warmup', '1', '--sweeps', '40'], capture_output=True, text=True, timeout=900, env=full,
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-3000:])
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['gpus_requested'] == 2 and d['collective_backend'] == 'gloo' and d['rccl_ranks'] == 0
    assert d['scaling'] == 'strong' and d['config']['members_total'] == 4 and d['config']['members_this_gpu'] == 2
    assert d['value'] > 0


def test_bench_c5_two_ranks_equal_one_rank():
    """The C5 strong-scaling split (`--config c5 --members 4 --gpus 2`, two ranks sharing this GPU over gloo): the
    loop indices / overflow flags and the per-volume checksums of the final S gathered from the two blocks are, bit for bit,
    those of a single rank solving all four volumes (the grouping of the norm's partial sums follows the launch's tiling --
    which tiles are cut into k chunks depends on the tile count: flags[:, 1] agrees to 1e-12, not hashed); the line
    carries every rank's own rate and rank 0's stand-alone rate (what the driver's N = 1 run is compared with)."""
    import json
    import os
    import subprocess
    import sys
    full = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        full.pop(k, None)
    full.update(XINV_FORCE_DEVICE='0', XINV_DIST_BACKEND='gloo')
    got = {}
    for ng in (2, 1):
        out = subprocess.run([sys.executable, 'bench.py', '--config', 'c5', '--members', '4', '--gpus', str(ng), '--steps', '1',
                              '--warmup', '0', '--sweeps', '11'], capture_output=True, text=True, timeout=1500, env=full,
                             cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-3000:])
        got[ng] = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][0])
    two, one = got[2], got[1]
    assert two['n_gpus'] == 2 and two['scaling'] == 'strong' and two['config']['members_total'] == 4
    assert two['config']['members_this_gpu'] == 2 and one['config']['members_this_gpu'] == 4
    assert two['flags_sha256'] == one['flags_sha256']                    # overflow flags and loop indices
    assert two['S_checksum_sha256'] == one['S_checksum_sha256']          # every volume's final S, bit for bit
    assert [r['members'] for r in two['rank_values']] == [2, 2] and all(r['value'] > 0 for r in two['rank_values'])
    assert two['n1_value'] > 0 and one['n1_value'] is None and one['rank_values'] is None


@pytest.mark.parametrize('cfg,members,grid', [('c4', 64, '144,288'), ('c5', 120, '50,72,144')])
def test_bench_eight_way_shard_shapes(cfg, members, grid):
    """The REAL 8-way splits of the batched BASELINE configurations -- C4: 64 members -> 8 per rank, C5: 120 volumes -> 15
    per rank (8.09 rounds of k_pipe3d's workgroups: the cut tail) -- as eight ranks sharing this GPU over gloo, on a reduced
    grid: every slice's loop index / overflow flag and the checksum of its final field are, bit for bit, those of one rank
    solving the whole batch (reference core.py:129: no cross-slice state)."""
    import json
    import os
    import subprocess
    import sys
    full = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        full.pop(k, None)
    full.update(XINV_FORCE_DEVICE='0', XINV_DIST_BACKEND='gloo')
    got = {}
    for ng in (8, 1):
        out = subprocess.run([sys.executable, 'bench.py', '--config', cfg, '--members', str(members), '--grid', grid, '--gpus', str(ng),
                              '--steps', '1', '--warmup', '0', '--sweeps', '21'], capture_output=True, text=True, timeout=1500,
                             env=full, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-3000:])
        got[ng] = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][0])
    eight, one = got[8], got[1]
    assert eight['n_gpus'] == 8 and eight['scaling'] == 'strong' and eight['config']['members_total'] == members
    assert [r['members'] for r in eight['rank_values']] == [members // 8] * 8
    assert eight['flags_sha256'] == one['flags_sha256']
    assert eight['S_checksum_sha256'] == one['S_checksum_sha256']
    assert eight['rank_spread'] >= 1.0 and eight['n1_value'] > 0


def test_bench_refuses_more_ranks_than_gpus():
    import os
    import subprocess
    import sys
    from xinvert_amd import _lib
    ndev = _lib.load().xinv_device_count()
    full = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'XINV_FORCE_DEVICE'):
        full.pop(k, None)
    out = subprocess.run([sys.executable, 'bench.py', '--gpus', str(ndev + 1), '--steps', '1', '--warmup', '0'],
                         capture_output=True, text=True, timeout=600, env=full,
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.returncode != 0 and 'refusing to run fewer ranks than asked' in (out.stderr + out.stdout)


def test_bench_single_rank_through_rccl():
    """bench.py's N > 1 code path -- barrier, flag all-gather, max-over-ranks timing -- over RCCL on this GPU:
    one rank joined as a process group (XINV_DIST_FORCE_INIT=1); the JSON line counts the ranks RCCL saw."""
    import json
    import os
    import socket
    import subprocess
    import sys
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    full = dict(os.environ)
    full.update(XINV_DIST_FORCE_INIT='1', RANK='0', LOCAL_RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1',
                MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    out = subprocess.run([sys.executable, 'bench.py', '--ny', '360', '--nx', '720', '--steps', '2', '--warmup', '1',
                          '--sweeps', '40', '--no-cpu', '--no-hbm', '--no-configs', '--no-parity'],
                         capture_output=True, text=True, timeout=900, env=full,
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-3000:])
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][0])
    assert d['collective_backend'] == 'nccl' and d['rccl_ranks'] == 1 and d['n_gpus'] == 1
