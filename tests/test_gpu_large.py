"""Batches of more than 2^31 elements per array: 64-bit offsets on every path.

The oracle cannot run these sizes, so the check is a size-independent property: slices of a batch
are independent (reference core.py:129), hence a member that lies entirely beyond element 2^31
of the batch must come out bit for bit like the same data solved alone.  Data are a small random
problem tiled over the big grid on the device (nothing this size crosses PCIe).
"""
import ctypes

import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu

NB = 10            # 10 x 2^28 elements: members 8 and 9 start at / beyond element 2^31


def _need_hbm(gib):
    import torch
    free, _ = torch.cuda.mem_get_info()
    if free < gib * 2**30:
        pytest.skip('needs %d GiB of free HBM' % gib)


def _tile(a, reps):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).cuda().repeat(*reps).contiguous()


def _solve_dev(p, ts, strides, nb, sweeps, **opt):
    from xinvert_amd import _lib
    L = _lib.require_gpu()
    fl = np.tile(np.array([0., 1., 0.]), (nb, 1))
    o = _lib.options(**opt)
    rc = getattr(L, util._FN[p['kind']] + '_dev')(
        *[ctypes.c_void_p(t.data_ptr()) if t is not None else None for t in ts], nb,
        _lib.strides_arg(strides), *util._scal(p, fl, sweeps - 1, 0.0), ctypes.byref(o), None)
    _lib.check(rc)
    return fl, _lib.last_stats()


def _big_problem(kind, BCy, BCx):
    """Two small problems (even / odd members) and how often to repeat them per axis."""
    if kind in ('std3d', 'gen3d'):
        mk = util.rand3d if kind == 'std3d' else util.rand3dg
        ps = [mk(8, 128, 128, BCy, BCx, msk=True, seed=s) for s in (11, 12)]
        reps = (8, 16, 16)                                  # 64 x 2048 x 2048 = 2^28
    elif kind == 'bih2d':
        ps = [util.randbih(256, 256, BCy, BCx, msk=True, seed=s) for s in (11, 12)]
        reps = (64, 64)                                     # 16384 x 16384 = 2^28
    elif kind == 'std2dt':
        ps = [util.rand2dt(256, 256, BCy, BCx, msk=True, seed=s) for s in (11, 12)]
        reps = (64, 64)
    else:
        ps = [util.rand2d(kind, 256, 256, BCy, BCx, bnz=(kind == 'gen2d'), msk=True, seed=s)
              for s in (11, 12)]
        reps = (64, 64)
    return ps, reps


CASES = [('std2d', 'fixed', 'periodic', {}),
         ('std2d', 'extend', 'fixed', dict(no_tile_skip=1)),
         ('gen2d', 'fixed', 'fixed', {}),
         ('std2dt', 'fixed', 'periodic', {}),
         ('bih2d', 'fixed', 'fixed', {}),
         ('std3d', 'fixed', 'periodic', {}),
         ('gen3d', 'extend', 'fixed', {}),
         ('std2d', 'fixed', 'fixed', dict(path=1)),
         ('std3d', 'fixed', 'fixed', dict(path=1))]


@pytest.mark.parametrize('kind,BCy,BCx,opt', CASES)
def test_members_beyond_2G_elements(kind, BCy, BCx, opt):
    import torch
    _need_hbm(120)
    ps, reps = _big_problem(kind, BCy, BCx)
    p = dict(ps[0])
    shape = tuple(int(s * r) for s, r in zip(ps[0]['S0'].shape, reps))
    n = int(np.prod(shape))
    assert NB * n > 2**31 and 8 * n >= 2**31
    if len(shape) == 3:
        p['zc'], p['yc'], p['xc'] = shape
    else:
        p['yc'], p['xc'] = shape
    nco = len(p['coefs'])
    per_member = {0, nco - 1}               # first coefficient and the forcing vary per member
    # a band of fully undefined forcing rows so that masked-tile skipping has something to skip
    band = slice(shape[-2] // 4, shape[-2] // 4 + shape[-2] // 8)

    def big(q, k):
        t = _tile(q['coefs'][k], reps)
        if k == nco - 1:
            t[..., band, :] = util.U
        return t

    S = torch.empty((NB,) + shape, dtype=torch.float64, device='cuda')
    for m in range(NB):
        S[m] = _tile(ps[m % 2]['S0'], reps)
    ts, strides = [S], [n]
    for k in range(nco):
        if k in per_member:
            t = torch.empty((NB,) + shape, dtype=torch.float64, device='cuda')
            e, o = big(ps[0], k), big(ps[1], k)
            for m in range(NB):
                t[m] = o if m % 2 else e
            del e, o
            ts.append(t); strides.append(n)
        else:
            ts.append(big(ps[0], k)); strides.append(0)
    torch.cuda.synchronize()
    # members 8 (even data) and 9 (odd data), each alone, from untouched copies
    alone = []
    for m in (NB - 2, NB - 1):
        one = [S[m].clone()] + [t[m] if st else t for t, st in zip(ts[1:], strides[1:])]
        fl1, _ = _solve_dev(p, one, strides, 1, 6, **opt)
        alone.append((one[0], fl1[0].copy()))
    fl, st = _solve_dev(p, ts, strides, NB, 6, **opt)
    torch.cuda.synchronize()
    for (S1, f1), m in zip(alone, (NB - 2, NB - 1)):
        assert torch.equal(S[m], S1), '%s: member %d differs from the same data solved alone' % (kind, m)
        assert fl[m][2] == f1[2] == 5 and fl[m][0] == f1[0]
        assert abs(fl[m][1] - f1[1]) <= 1e-12
    # identical data => identical result wherever the member sits in the batch
    assert torch.equal(S[0], S[NB - 2]) and torch.equal(S[1], S[NB - 1])
    assert np.array_equal(fl[0::2, [0, 2]], np.tile(fl[0, [0, 2]], (NB // 2, 1)))
    assert np.allclose(fl[0::2, 1], fl[0, 1], rtol=0, atol=1e-12)
    assert bool(torch.isfinite(S[NB - 1][S[NB - 1] != util.U]).all())


@pytest.mark.parametrize('BCy', ['fixed', 'extend'])
@pytest.mark.parametrize('sweeps,tol', [(24, 0.0), (25, 0.0), (400, 3e-3)])
@pytest.mark.parametrize('uni', [0, 1])
def test_rolling_batch_3d_host_pointers(BCy, sweeps, tol, uni):
    """The host-pointer entry of the standard 3-D form with shared coefficient arrays runs a ROLLING batch (xinv_hostptr.h:
    roll_3d): one chain of launches over the volumes that have arrived and still have sweeps to do -- a volume joins at an
    even launch, retires after its budget (an odd budget: a last one-sweep launch for the retiring group beside the
    others' two-sweep launch), a volume the tolerance stops earlier idles, one that stops INSIDE a pass is redone from that
    pass's source.  Bit for bit the oracle per volume, flags included; `host_inflight = -1` takes the path at this size."""
    from util import rand3d
    nb = 7
    base = rand3d(12, 40, 136, BCy, 'periodic', 1, seed=900)
    if uni:
        for q in range(3):
            base['coefs'][q][:] = base['coefs'][q][:, :, :1]
    ps = []
    for m in range(nb):
        q = dict(base)
        r = rand3d(12, 40, 136, BCy, 'periodic', 1, seed=901 + m)
        q['coefs'] = list(base['coefs'][:3]) + [r['coefs'][3]]
        q['S0'] = r['S0']
        ps.append(q)
    S, fl, st = util.run_hip_batched(ps, sweeps - 1, tol, shared=(0, 1, 2), host_inflight=-1)
    assert st['path'] == 2 and st['host_chunks'] == nb and st['rolling'] == 1, st
    loops = set()
    for m, q in enumerate(ps):
        So, flo = util.run_oracle(q, sweeps - 1, tol, 2)
        assert np.array_equal(S[m], So), 'member %d: %d points differ' % (m, (S[m] != So).sum())
        assert fl[m][0] == flo[0] and fl[m][2] == flo[2] and abs(fl[m][1] - flo[1]) <= 1e-12 * max(1.0, abs(flo[1])), (m, fl[m], flo)
        loops.add(int(flo[2]))
    if tol > 0:
        assert len(loops) > 1 and {l % 2 for l in loops} == {0, 1}, loops      # (stops at both parities: inside and at the end of a pass)
    # the chunked pipeline gives the same fields
    S2, fl2, st2 = util.run_hip_batched(ps, sweeps - 1, tol, shared=(0, 1, 2), host_chunk=2)
    assert np.array_equal(S, S2) and np.array_equal(fl[:, [0, 2]], fl2[:, [0, 2]])


@pytest.mark.parametrize('kind', ['std2d', 'gen2d'])
@pytest.mark.parametrize('sweeps,tol', [(40, 0.0), (41, 0.0), (600, 2e-3)])
@pytest.mark.parametrize('masked_tiles', [0, 1])
def test_rolling_batch_2d_host_pointers(kind, sweeps, tol, masked_tiles):
    """The rolling batch for the 2-D 5-point forms with shared per-row coefficients (lat-lon Poisson / Gill-Matsuno batches):
    the first chunk is planned alone; if it has no fully masked tile the batch is planned without tile lists and rolls (four
    sweeps per pass on k_pipe2d, a shorter tail launch for a budget that is not whole passes, stops at any sweep of a pass);
    if it has, the chunk scheme takes the call.  Bit for bit the oracle either way."""
    from util import rand2d
    nb = 6
    yc, xc = 150, 460
    base = rand2d(kind, yc, xc, 'fixed', 'periodic', 0, 0, seed=950)
    ncu = 3 if kind == 'std2d' else 6
    for q in range(ncu):
        base['coefs'][q][:] = base['coefs'][q][:, :1]
    ps = []
    for m in range(nb):
        r = rand2d(kind, yc, xc, 'fixed', 'periodic', 0, 1, seed=951 + m)
        q = dict(base)
        F = r['coefs'][-1] * 10.0 ** (-(m % 3))
        F[r['coefs'][-1] == util.U] = util.U
        if masked_tiles:
            F[:, :130] = util.U                          # a strip of land: whole tiles to skip
        q['coefs'] = list(base['coefs'][:ncu]) + [F]
        q['S0'] = np.where(r['S0'] == util.U, 0.0, r['S0'])
        ps.append(q)
    ref = [util.run_oracle(q, sweeps - 1, tol, 2) for q in ps]
    # one chunk (a batch this small): one rolling chain; chunks of two / of one member: TWO chains -- the lanes of the
    # rolling batch, members 0-3 / 4-5 and 0-2 / 3-5 --, their launches issued alternately (xinv_hostptr.h)
    for host_chunk in (0, 2, 1):
        S, fl, st = util.run_hip_batched(ps, sweeps - 1, tol, shared=tuple(range(ncu)), host_inflight=-1, force_tile_skip=1,
                                         host_chunk=host_chunk)
        assert st['path'] == 2 and st['rolling'] == (0 if masked_tiles else 1), st
        if not masked_tiles:
            assert st['lanes'] == (2 if host_chunk else 1) and st['host_chunks'] == (nb // host_chunk if host_chunk else 1), st
        for m, q in enumerate(ps):
            So, flo = ref[m]
            assert np.array_equal(S[m], So), 'chunks of %d, member %d: %d points differ' % (host_chunk, m, (S[m] != So).sum())
            assert fl[m][0] == flo[0] and fl[m][2] == flo[2], (host_chunk, m, fl[m], flo)
