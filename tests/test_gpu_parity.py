"""GPU parity: the HIP path (through the C-ABI) against the coloured-ordering oracle.

Bar: S bit-exact (same ordering, same arithmetic, no FMA contraction on either side);
flags[2] (loop index) equal; flags[1] = |norm - normPrev| / normPrev to 1e-12 ABSOLUTE: the
device adds the mean|S| partials in a different (fixed) order than the serial oracle, which
perturbs each norm by a few ulp (~1e-16 relative) and therefore their relative DIFFERENCE by
~1e-15 absolute, whatever its size.
"""
import zlib

import ctypes

import numpy as np
import pytest

import util
from util import rand2d, rand2dt, rand3d, rand3dg, randbih, run_oracle, run_hip_single, run_hip_batched, run_hip_dev

pytestmark = pytest.mark.gpu

def _seed(key):
    return zlib.crc32(repr(key).encode()) & 0x7fffffff


COLOUR_AUTO, COLOUR_2 = 1, 2
PATH_COLOUR, PATH_FUSED = 1, 2


def assert_same(S, fl, So, flo, what=''):
    if not np.array_equal(S, So):
        d = (S != So)
        axes = [np.where(d.any(axis=tuple(a for a in range(d.ndim) if a != k)))[0] for k in range(d.ndim)]
        where = ' x '.join('%d..%d (%d)' % (a.min(), a.max(), len(a)) for a in axes)
        raise AssertionError('%s: S differs, max |d| = %g at %d points, index ranges %s' % (
            what, np.nanmax(np.abs(S - So)), int(d.sum()), where))
    assert fl[2] == flo[2], '%s: loop index %r vs oracle %r' % (what, fl[2], flo[2])
    assert fl[0] == flo[0]
    if np.isnan(flo[1]):
        assert np.isnan(fl[1]), (what, fl, flo)
    else:
        assert abs(fl[1] - flo[1]) <= 1e-12 + 1e-9 * abs(flo[1]), (what, fl, flo)


BCS = [('fixed', 'fixed'), ('fixed', 'periodic'), ('extend', 'fixed'), ('extend', 'periodic'),
       ('fixed', 'extend')]


@pytest.mark.parametrize('kind', ['std2d', 'gen2d'])
@pytest.mark.parametrize('BCy,BCx', BCS)
@pytest.mark.parametrize('bnz', [0, 1])
@pytest.mark.parametrize('msk', [0, 1])
@pytest.mark.parametrize('shape', [(17, 24), (12, 19), (21, 70)])
def test_colour_path_2d(kind, BCy, BCx, bnz, msk, shape):
    p = rand2d(kind, shape[0], shape[1], BCy, BCx, bnz, msk, seed=_seed((kind, BCy, BCx, bnz, msk, shape)))
    So, flo = run_oracle(p, 25, 1e-9, COLOUR_AUTO)
    S, fl, st = run_hip_batched([p], 25, 1e-9, path=PATH_COLOUR)
    assert st['path'] == PATH_COLOUR
    assert_same(S[0], fl[0], So, flo, 'colour path %s' % kind)


@pytest.mark.parametrize('BCy,BCx', [('fixed', 'fixed'), ('fixed', 'periodic'), ('extend', 'periodic'), ('extend', 'fixed')])
@pytest.mark.parametrize('msk', [0, 1])
@pytest.mark.parametrize('shape', [(6, 9, 12), (5, 7, 9), (7, 10, 66)])
def test_colour_path_3d(BCy, BCx, msk, shape):
    p = rand3d(shape[0], shape[1], shape[2], BCy, BCx, msk, seed=_seed((BCy, BCx, msk, shape)))
    So, flo = run_oracle(p, 15, 1e-9, COLOUR_AUTO)
    S, fl, st = run_hip_batched([p], 15, 1e-9)
    assert_same(S[0], fl[0], So, flo, '3d')


@pytest.mark.parametrize('BCy,BCx', [('fixed', 'fixed'), ('fixed', 'periodic'), ('extend', 'periodic'),
                                     ('extend', 'fixed'), ('fixed', 'extend')])
@pytest.mark.parametrize('msk', [0, 1])
@pytest.mark.parametrize('shape', [(6, 9, 12), (5, 7, 9), (7, 10, 66), (4, 12, 8), (3, 3, 3)])
def test_general_3d_colour_path(BCy, BCx, msk, shape):
    """numbas.invert_general_3D: red-black on (k+j+i)&1 (+2 seam colours for odd-xc periodic),
    the west-periodic branch that never tests H, the pre-pass with its range(1, yc-1) second
    loop (yc > xc), single-slice and batched entries."""
    p = rand3dg(shape[0], shape[1], shape[2], BCy, BCx, msk, seed=_seed(('g3', BCy, BCx, msk, shape)))
    So, flo = run_oracle(p, 15, 1e-9, COLOUR_AUTO)
    S, fl, st = run_hip_batched([p, p], 15, 1e-9)
    assert st['path'] == PATH_COLOUR
    assert st['colours'] == 2 + (2 if BCx == 'periodic' and shape[2] % 2 else 0)
    assert_same(S[0], fl[0], So, flo, 'gen3d')
    assert np.array_equal(S[0], S[1])
    S1, f1 = run_hip_single(p, 15, 1e-9)
    assert np.array_equal(S1, S[0]) and np.array_equal(f1, fl[0])


def test_general_3d_dev_and_early_stop():
    ps = [rand3dg(6, 14, 40, 'fixed', 'periodic', 1, seed=s) for s in (1, 2, 3)]
    S1, f1, _ = run_hip_batched(ps, 400, 1e-5)
    S2, f2, _ = run_hip_dev(ps, 400, 1e-5)
    assert np.array_equal(S1, S2) and np.array_equal(f1, f2)
    loops = set()
    for m, q in enumerate(ps):
        So, flo = run_oracle(q, 400, 1e-5, COLOUR_2)
        assert_same(S1[m], f1[m], So, flo, 'gen3d member %d' % m)
        loops.add(flo[2])
    assert len(loops) > 1


def _uniform3dg(p):
    """A..G constant along x (every 3DOcean coefficient is a function of level and latitude)."""
    q = dict(p)
    q['coefs'] = [np.ascontiguousarray(np.broadcast_to(c[..., :1], c.shape)) if k < 7 else c
                  for k, c in enumerate(p['coefs'])]
    return q


@pytest.mark.parametrize('BCy,BCx', [('fixed', 'fixed'), ('fixed', 'periodic'), ('extend', 'periodic'),
                                     ('extend', 'fixed'), ('fixed', 'extend')])
@pytest.mark.parametrize('msk', [0, 1])
@pytest.mark.parametrize('nw', [0, 8, 16])
@pytest.mark.parametrize('shape', [(6, 9, 12), (5, 30, 130), (9, 21, 260), (4, 14, 10), (3, 3, 4), (70, 9, 12),
                                   (45, 30, 130)])
def test_general_3d_fused_path(BCy, BCx, msk, nw, shape):
    """k_fused3dg: the general 3-D form on the streaming path when A..G are constant along x --
    red-black order of the oracle bit for bit, including the west-periodic branch that never tests H,
    the pre-pass with its range(1, yc-1) second loop, k chunks on tall volumes, batches."""
    ps = [_uniform3dg(rand3dg(shape[0], shape[1], shape[2], BCy, BCx, msk, seed=_seed(('g3f', BCy, BCx, msk, shape, m))))
          for m in range(2)]
    S, fl, st = run_hip_batched(ps, 14, 1e-9, path=PATH_FUSED, rows_per_tile=nw)
    assert st['path'] == PATH_FUSED and st['xuniform_mask'] == 0x7f
    assert st['rows_per_tile'] == (nw if nw == 8 else 12)   # (sixteen wavefronts spilled: not instantiated, twelve run)
    for m, q in enumerate(ps):
        So, flo = run_oracle(q, 14, 1e-9, COLOUR_2)
        assert_same(S[m], fl[m], So, flo, 'gen3d fused %r member %d' % (shape, m))
    # coefficients that vary along x have no fused kernel
    with pytest.raises(Exception, match='no fused kernel'):
        run_hip_batched([rand3dg(shape[0], shape[1], max(shape[2], 4), BCy, BCx, msk, seed=3)], 3, 0.0, path=PATH_FUSED)


def _uniform3d(p, rng):
    """Make A, B, C constant along x (what every lat-lon omega coefficient looks like)."""
    q = dict(p)
    q['coefs'] = [np.ascontiguousarray(np.broadcast_to(c[..., :1], c.shape)) if k < 3 else c
                  for k, c in enumerate(p['coefs'])]
    return q


@pytest.mark.parametrize('BCy,BCx', [('fixed', 'fixed'), ('fixed', 'periodic'), ('extend', 'periodic'), ('extend', 'fixed')])
@pytest.mark.parametrize('msk', [0, 1])
@pytest.mark.parametrize('uni', [0, 1])
@pytest.mark.parametrize('nw', [8, 12, 16])
@pytest.mark.parametrize('shape', [(6, 9, 12), (5, 30, 130), (9, 21, 260), (4, 14, 11), (3, 3, 3)])
def test_fused_path_3d(BCy, BCx, msk, uni, nw, shape):
    if BCx == 'periodic' and shape[2] % 2:
        pytest.skip('odd-xc periodic seam on rows shorter than 64 columns goes through the colour path (longer rows: test_gpu_seam.py)')
    p = rand3d(shape[0], shape[1], shape[2], BCy, BCx, msk, seed=_seed((BCy, BCx, msk, uni, shape)))
    if uni:
        p = _uniform3d(p, None)
    So, flo = run_oracle(p, 12, 1e-9, COLOUR_2)
    S, fl, st = run_hip_batched([p], 12, 1e-9, path=PATH_FUSED, rows_per_tile=nw)
    # (sixteen wavefronts = 128 VGPRs per lane: only the x-uniform variant without 'extend' fits; the others get twelve)
    nw_ran = 12 if (nw == 16 and not (uni and BCy != 'extend')) else nw
    assert st['path'] == PATH_FUSED and st['rows_per_tile'] == nw_ran
    assert st['xuniform_mask'] == (7 if uni else 0)
    assert_same(S[0], fl[0], So, flo, 'fused 3d %r' % (shape,))


@pytest.mark.parametrize('BCy,BCx', [('fixed', 'fixed'), ('fixed', 'periodic'), ('extend', 'periodic')])
@pytest.mark.parametrize('uni', [0, 1])
@pytest.mark.parametrize('shape', [(70, 9, 12), (45, 30, 130), (33, 14, 260), (129, 5, 8), (64, 12, 20)])
def test_fused_path_3d_k_chunks(BCy, BCx, uni, shape):
    """Tall volumes are split into k chunks (two halo planes a side, recomputed): every chunk
    boundary must be invisible -- bit for bit the oracle, masks and early stop included."""
    ps = [rand3d(shape[0], shape[1], shape[2], BCy, BCx, 1, seed=_seed(('kc', BCy, BCx, uni, shape, m)))
          for m in range(2)]
    if uni:
        ps = [_uniform3d(q, None) for q in ps]
    S, fl, st = run_hip_batched(ps, 40, 1e-4, path=PATH_FUSED)
    assert st['path'] == PATH_FUSED
    for m, q in enumerate(ps):
        So, flo = run_oracle(q, 40, 1e-4, COLOUR_2)
        assert_same(S[m], fl[m], So, flo, 'k-chunks %r member %d' % (shape, m))


def test_fused_3d_batched_early_stop():
    ps = [rand3d(7, 20, 140, 'fixed', 'periodic', 1, seed=s) for s in (1, 2, 3)]
    S, fl, st = run_hip_batched(ps, 300, 2e-4)
    assert st['path'] == PATH_FUSED
    loops = set()
    for m, q in enumerate(ps):
        So, flo = run_oracle(q, 300, 2e-4, COLOUR_2)
        assert_same(S[m], fl[m], So, flo, '3d member %d' % m)
        loops.add(flo[2])
    assert len(loops) > 1


FUSED_SHAPES = [(17, 24), (12, 20), (40, 300), (70, 130), (33, 257), (8, 4), (3, 3), (64, 512)]


@pytest.mark.parametrize('kind', ['std2d', 'gen2d'])
@pytest.mark.parametrize('BCy,BCx', BCS)
@pytest.mark.parametrize('msk', [0, 1])
@pytest.mark.parametrize('shape', FUSED_SHAPES)
@pytest.mark.parametrize('K', [1, 2, 3, 4])
def test_fused_path(kind, BCy, BCx, msk, shape, K):
    yc, xc = shape
    if BCx == 'periodic' and xc % 2 and xc < 64:
        pytest.skip('odd-xc periodic seam on rows shorter than 64 columns goes through the colour path (covered there)')
    p = rand2d(kind, yc, xc, BCy, BCx, 0, msk, seed=_seed((kind, BCy, BCx, msk, shape)))
    So, flo = run_oracle(p, 24, 1e-9, COLOUR_2)
    S, fl, st = run_hip_batched([p], 24, 1e-9, path=PATH_FUSED, sweeps_per_launch=K, rows_per_tile=16)
    # the general form has K = 1, 2, 3; the standard form also 4
    assert st['path'] == PATH_FUSED and st['sweeps_per_launch'] == (K if kind == 'std2d' else min(K, 3))
    assert_same(S[0], fl[0], So, flo, 'fused K=%d %s %r' % (K, kind, shape))


@pytest.mark.parametrize('kind', ['std2d', 'gen2d'])
@pytest.mark.parametrize('K', [1, 2, 3])
@pytest.mark.parametrize('tol', [3e-3, 1e-3, 2e-4])
def test_fused_early_stop_exact_sweep(kind, K, tol):
    """Stopping inside a K-sweep launch must return the state of exactly the stopping sweep."""
    p = rand2d(kind, 40, 300, 'fixed', 'periodic', 0, 1, seed=7)
    So, flo = run_oracle(p, 500, tol, COLOUR_2)
    assert 2 < flo[2] < 499
    S, fl, st = run_hip_batched([p], 500, tol, path=PATH_FUSED, sweeps_per_launch=K, rows_per_tile=8,
                                check_every=5)
    assert_same(S[0], fl[0], So, flo, 'early stop')
    assert st['sweeps_max'] == flo[2] + 1


def _uniform2d(p):
    """A and C constant along x (a lat-lon grid's cos(lat) factors): the per-row-scalar variants."""
    q = dict(p); cs = [np.array(c, copy=True) for c in p['coefs']]
    ic = 2                                            # std2d: A B C F ; gen2d: A B C D E F G
    cs[0] = np.repeat(cs[0][:, :1], cs[0].shape[1], axis=1)
    cs[ic] = np.repeat(cs[ic][:, :1], cs[ic].shape[1], axis=1)
    q['coefs'] = cs
    return q


@pytest.mark.parametrize('BCy,BCx', BCS)
@pytest.mark.parametrize('msk', [0, 1])
@pytest.mark.parametrize('shape', FUSED_SHAPES + [(90, 250)])
@pytest.mark.parametrize('K', [3, 4])
@pytest.mark.parametrize('nsw', [24, 22])
def test_fused_standard_three_and_four_sweeps_per_pass(BCy, BCx, msk, shape, K, nsw):
    """K = 3, 4 (standard form with per-row A, C): same ordering, so bit for bit the oracle;
    nsw + 1 sweeps leave a tail of 1 (25 = 6x4 + 1 = 8x3 + 1) or 3 / 2 (23) for a shorter last pass."""
    yc, xc = shape
    if BCx == 'periodic' and xc % 2 and xc < 64:
        pytest.skip('odd-xc periodic seam on rows shorter than 64 columns goes through the colour path (covered there)')
    p = _uniform2d(rand2d('std2d', yc, xc, BCy, BCx, 0, msk, seed=_seed(('k34', BCy, BCx, msk, shape))))
    So, flo = run_oracle(p, nsw, 1e-9, COLOUR_2)
    for rows in (16, 0):
        S, fl, st = run_hip_batched([p], nsw, 1e-9, path=PATH_FUSED, sweeps_per_launch=K, rows_per_tile=rows)
        assert st['path'] == PATH_FUSED and st['sweeps_per_launch'] == K, st
        assert_same(S[0], fl[0], So, flo, 'fused K=%d %r rows=%d' % (K, shape, rows))
    S, fl, st = run_hip_batched([p], nsw, 1e-9, path=PATH_FUSED)            # auto: the deepest variant
    assert st['sweeps_per_launch'] in (3, 4)
    assert_same(S[0], fl[0], So, flo, 'fused K=auto %r' % (shape,))


def _uniform2d_all(p):
    """every coefficient array but the forcing constant along x (lat-lon Gill-Matsuno: A, C, D, E, F per row)"""
    q = dict(p); cs = [np.array(c, copy=True) for c in p['coefs']]
    for k in range(len(cs) - 1):
        if k != 1:                                    # (B stays what it is: identically zero here)
            cs[k] = np.repeat(cs[k][:, :1], cs[k].shape[1], axis=1)
    q['coefs'] = cs
    return q


@pytest.mark.parametrize('BCy,BCx', [('fixed', 'fixed'), ('fixed', 'periodic'), ('extend', 'periodic'), ('extend', 'fixed')])
@pytest.mark.parametrize('shape', [(40, 300), (33, 257), (64, 512), (90, 250), (200, 1200)])
@pytest.mark.parametrize('kind', ['std2d', 'gen2d'])
def test_pipelined_pass_equals_single_wavefront_pass(BCy, BCx, shape, kind):
    """k_pipe2d (four sweeps pipelined across the wavefronts of a workgroup; standard form with per-row A, C --
    and general form with per-row A, C, D, E, F) against k_fused2d
    (XINV_FLAG_NO_PIPE) and the oracle: bit for bit, masked tiles skipped or not, several members with
    their own coefficients, stats.pipelined reporting which kernel ran."""
    import os
    yc, xc = shape
    seam = BCx == 'periodic' and xc % 2 == 1          # (the seam variants hold one column pair per lane)
    uni = {'std2d': _uniform2d, 'gen2d': _uniform2d_all}[kind]          # per-row A, C / A, C, D, E, F
    um = {'std2d': 3, 'gen2d': 31}[kind]
    ps = [uni(rand2d(kind, yc, xc, BCy, BCx, 0, m & 1, seed=_seed(('pipe', kind, BCy, BCx, shape, m)))) for m in range(3)]
    ref = [run_oracle(p, 26, 1e-9, COLOUR_2) for p in ps]
    S0, f0, st0 = run_hip_batched(ps, 26, 1e-9, path=PATH_FUSED, sweeps_per_launch=4, no_pipe=1)
    assert st0['pipelined'] == 0 and st0['xuniform_mask'] == um, st0
    # one / two column pairs per lane; the forcing re-read from memory by every wavefront / riding the LDS ring
    # the forcing re-read from memory by every wavefront / riding the LDS ring (xinv_options.pipe_fr = -1 / 1)
    for fr in (-1, 1):
        for kw in (dict(), dict(rows_per_tile=16), dict(force_tile_skip=1), dict(rows_per_tile=-3), dict(sweeps_per_launch=4)):
            o = dict(path=PATH_FUSED, pipe_fr=fr); o.update(kw)
            S, fl, st = run_hip_batched(ps, 26, 1e-9, **o)
            assert st['pipelined'] == 1 and st['sweeps_per_launch'] == 4, st
            for m in range(3):
                assert_same(S[m], fl[m], ref[m][0], ref[m][1], 'pipelined %s %r member %d %r fr %d' % (kind, shape, m, kw, fr))
            assert np.array_equal(S, S0)


@pytest.mark.parametrize('tol', [3e-3, 1e-3, 2e-4, 5e-5])
def test_pipelined_general_form_early_stop_exact_sweep(tol):
    """The stop rule firing inside a four-sweep pipelined pass of the general form returns exactly the oracle's
    stopping sweep (redo from the pass's source with the single-sweep kernel), lagged norm or not."""
    p = _uniform2d_all(rand2d('gen2d', 40, 300, 'fixed', 'periodic', 0, 1, seed=7))
    So, flo = run_oracle(p, 500, tol, COLOUR_2)
    assert 2 < flo[2] < 499
    for kw in (dict(rows_per_tile=8, check_every=5), dict(), dict(force_tile_skip=1)):
        S, fl, st = run_hip_batched([p], 500, tol, path=PATH_FUSED, **kw)
        assert st['pipelined'] == 1 and st['sweeps_per_launch'] == 4, st
        assert_same(S[0], fl[0], So, flo, 'general form, early stop %r' % (kw,))
        assert st['sweeps_max'] == flo[2] + 1


@pytest.mark.parametrize('K', [3, 4])
@pytest.mark.parametrize('tol', [3e-3, 1e-3, 2e-4, 5e-5])
def test_fused_early_stop_exact_sweep_deep_passes(K, tol):
    """Stopping inside a 3- or 4-sweep launch returns exactly the stopping sweep."""
    p = _uniform2d(rand2d('std2d', 40, 300, 'fixed', 'periodic', 0, 1, seed=7))
    So, flo = run_oracle(p, 500, tol, COLOUR_2)
    assert 2 < flo[2] < 499
    for skip in (0, 1):
        S, fl, st = run_hip_batched([p], 500, tol, path=PATH_FUSED, sweeps_per_launch=K, rows_per_tile=8,
                                    check_every=5, force_tile_skip=skip)
        assert st['sweeps_per_launch'] == K
        assert_same(S[0], fl[0], So, flo, 'early stop K=%d' % K)
        assert st['sweeps_max'] == flo[2] + 1


def test_fused_equals_colour_path():
    p = rand2d('gen2d', 50, 260, 'extend', 'periodic', 0, 1, seed=11)
    S1, f1, _ = run_hip_batched([p], 30, 0.0, path=PATH_FUSED)
    S2, f2, _ = run_hip_batched([p], 30, 0.0, path=PATH_COLOUR)
    assert np.array_equal(S1, S2)
    assert np.allclose(f1, f2, rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize('kind', ['std2d', 'gen2d'])
def test_batched_shared_coefficients_and_per_member_stop(kind):
    """Members stop at different sweeps; coefficients shared with batch stride 0."""
    base = rand2d(kind, 36, 200, 'fixed', 'periodic', 0, 0, seed=3)
    ps = []
    rng = np.random.default_rng(5)
    for m in range(5):
        q = dict(base)
        q['coefs'] = list(base['coefs'])
        q['coefs'][-1] = rng.standard_normal(base['S0'].shape) * (0.2 + m)
        q['S0'] = rng.standard_normal(base['S0'].shape) * 10.0 ** (m - 2)
        ps.append(q)
    shared = tuple(range(len(base['coefs']) - 1))
    S, fl, st = run_hip_batched(ps, 400, 5e-4, shared=shared, sweeps_per_launch=2)
    loops = set()
    for m, q in enumerate(ps):
        So, flo = run_oracle(q, 400, 5e-4, COLOUR_2)
        assert_same(S[m], fl[m], So, flo, 'member %d' % m)
        loops.add(flo[2])
    assert len(loops) > 1, 'test should exercise different stopping sweeps'


def _blocky(p, rng, boxes):
    """Mask whole rectangles of the forcing (land / topography): tiles inside them never change."""
    q = dict(p); q['coefs'] = [c.copy() for c in p['coefs']]; q['S0'] = p['S0'].copy()
    F = q['coefs'][-1]
    for (j0, j1, i0, i1) in boxes:
        F[j0:j1, i0:i1] = util.U
        q['S0'][j0:j1, i0:i1] = np.where(rng.random((j1 - j0, i1 - i0)) < 0.2, util.U, 0.25)
    return q


@pytest.mark.parametrize('kind', ['std2d', 'gen2d', 'std2dt'])
@pytest.mark.parametrize('BCy,BCx', [('fixed', 'fixed'), ('fixed', 'periodic'), ('extend', 'periodic'), ('extend', 'fixed')])
@pytest.mark.parametrize('spl', [1, 2, 3])
@pytest.mark.parametrize('rows', [-6, -15])
def test_masked_tiles_are_skipped_without_changing_a_bit(kind, BCy, BCx, spl, rows):
    """Masked-tile skipping (fully masked wave-tiles are left out of the launches, their constant
    share of the norm is added by the last workgroup): S bit for bit, flags, loop counts equal to
    the oracle -- with defined and undefined S inside the skipped tiles, members with different
    masks, 'extend' rows inside a masked block, an early stop inside a 2-sweep launch."""
    rng = np.random.default_rng(_seed(('skip', kind, BCy, BCx, spl, rows)))
    mk = (lambda s: rand2dt(60, 372, BCy, BCx, 0, 1, seed=s)) if kind == 'std2dt' else \
         (lambda s: rand2d(kind, 60, 372, BCy, BCx, 0, 1, seed=s))
    ps = [_blocky(mk(1), rng, [(0, 30, 0, 250), (44, 60, 120, 372)]),
          _blocky(mk(2), rng, [(10, 60, 100, 372)]),
          mk(3)]
    S, fl, st = run_hip_batched(ps, 60, 3e-5, path=PATH_FUSED, sweeps_per_launch=spl, rows_per_tile=rows,
                                force_tile_skip=1)
    assert st['path'] == PATH_FUSED and st['masked_tile_pct'] > 10
    S0, fl0, st0 = run_hip_batched(ps, 60, 3e-5, path=PATH_FUSED, sweeps_per_launch=spl, rows_per_tile=rows,
                                   no_tile_skip=1)
    assert st0['masked_tile_pct'] == 0
    loops = set()
    for m, q in enumerate(ps):
        So, flo = run_oracle(q, 60, 3e-5, COLOUR_2)
        assert_same(S[m], fl[m], So, flo, 'skip member %d' % m)
        assert np.array_equal(S[m], S0[m], equal_nan=True) and fl[m][2] == fl0[m][2]
        assert abs(fl[m][1] - fl0[m][1]) <= 1e-12
        loops.add(flo[2])
    assert max(loops) < 60                      # every member stopped on the tolerance


@pytest.mark.parametrize('kind,spl', [('std2d', 1), ('std2d', 2), ('std2d', 3), ('gen2d', 1), ('gen2d', 2)])
@pytest.mark.parametrize('BCy,BCx', [('fixed', 'fixed'), ('fixed', 'periodic'), ('extend', 'periodic')])
@pytest.mark.parametrize('rows', [-6, -12])
def test_masked_tiles_skipped_nine_point(kind, spl, BCy, BCx, rows):
    """The 4-colour fused kernel (B != 0) with masked-tile skipping: bit for bit the oracle and the
    run that visits every tile."""
    rng = np.random.default_rng(_seed(('skip9', kind, spl, BCy, BCx, rows)))
    ps = [_blocky(rand2d(kind, 60, 360, BCy, BCx, 1, 1, seed=11), rng, [(0, 30, 0, 250), (44, 60, 120, 360)]),
          _blocky(rand2d(kind, 60, 360, BCy, BCx, 1, 1, seed=12), rng, [(10, 60, 100, 360)])]
    S, fl, st = run_hip_batched(ps, 40, 1e-4, path=PATH_FUSED, sweeps_per_launch=spl, rows_per_tile=rows, force_tile_skip=1)
    assert st['path'] == PATH_FUSED and st['colours'] == 4 and st['masked_tile_pct'] > 0
    S0, fl0, st0 = run_hip_batched(ps, 40, 1e-4, path=PATH_FUSED, sweeps_per_launch=spl, rows_per_tile=rows, no_tile_skip=1)
    assert st0['masked_tile_pct'] == 0 and np.array_equal(S, S0) and np.array_equal(fl[:, 2], fl0[:, 2])
    for m, q in enumerate(ps):
        So, flo = run_oracle(q, 40, 1e-4, COLOUR_AUTO)
        assert_same(S[m], fl[m], So, flo, 'skip9 member %d' % m)


def test_masked_tile_skipping_degenerate_members():
    """A member whose forcing is masked everywhere (no active tile at all: the whole norm comes from
    the constant share), next to an ordinary one and to one with a NaN in a masked block's
    neighbourhood (NaN != undef: that tile must stay active and poison S as in the reference)."""
    rng = np.random.default_rng(77)
    a = rand2d('std2d', 48, 300, 'fixed', 'periodic', 0, 1, seed=5)
    dead = dict(a); dead['coefs'] = [c.copy() for c in a['coefs']]; dead['coefs'][-1][:] = util.U
    dead['S0'] = rng.standard_normal((48, 300)); dead['S0'][rng.random((48, 300)) < 0.1] = util.U
    nanm = _blocky(a, rng, [(0, 48, 0, 130)]); nanm['coefs'][-1][20, 60] = np.nan
    ps = [dead, a, nanm]
    S, fl, st = run_hip_batched(ps, 30, 1e-9, path=PATH_FUSED, rows_per_tile=-8, force_tile_skip=1)
    assert st['masked_tile_pct'] > 30
    for m, q in enumerate(ps):
        So, flo = run_oracle(q, 30, 1e-9, COLOUR_2)
        if m == 2:                      # S holds the NaN: compare NaN-aware
            assert np.array_equal(S[m], So, equal_nan=True) and np.isnan(S[m]).sum() >= 1
            assert fl[m][0] == flo[0] == 1.0
        else:
            assert_same(S[m], fl[m], So, flo, 'degenerate member %d' % m)
    assert np.array_equal(S[0], dead['S0']) and fl[0][2] == 1 and fl[0][1] == 0.0   # norm constant: stops at loop 1
    assert fl[2][0] == 1.0                                                          # NaN -> overflow exit


@pytest.mark.parametrize('kind', ['std2d', 'gen2d', 'std3d', 'bih2d'])
def test_row_constant_coefficients_travel_as_rows(kind):
    """xinv_options.rowconst_mask: coefficient arrays given as one value per row ([rows] per member,
    batch stride 0 or rows) are expanded on the device -- same bits as passing the full arrays."""
    from xinvert_amd import _lib
    L = _lib.require_gpu()
    if kind == 'std3d':
        ps = [rand3d(5, 12, 40, 'fixed', 'periodic', 1, seed=s) for s in (1, 2)]
    elif kind == 'bih2d':
        ps = [randbih(14, 40, 'fixed', 'fixed', 1, 1, seed=s) for s in (1, 2)]
    else:
        ps = [rand2d(kind, 14, 40, 'fixed', 'periodic', 0, 1, seed=s) for s in (1, 2)]
    nco = len(ps[0]['coefs'])
    rowq = [0, 2] if kind != 'bih2d' else [0, 2, 3, 5, 8]          # made constant along x
    ps[1]['coefs'][0] = ps[0]['coefs'][0]                           # array 0 is shared by the members
    for q in ps:
        q['coefs'] = [np.ascontiguousarray(np.broadcast_to(c[..., :1], c.shape)) if k in rowq else c
                      for k, c in enumerate(q['coefs'])]
    S_full, fl_full, _ = run_hip_batched(ps, 12, 1e-9, shared=(0,))
    # the same call with rows: array 0 shared (stride 0), the others per member (stride = rows)
    p = ps[0]; nb = len(ps); n = int(np.prod(p['S0'].shape)); rows = n // p['S0'].shape[-1]
    S = np.ascontiguousarray(np.stack([q['S0'] for q in ps]))
    arrs, strides, mask = [S], [n], 0
    for k in range(nco):
        if k in rowq:
            mask |= 1 << k
            if k == 0:
                arrs.append(np.ascontiguousarray(p['coefs'][k][..., 0])); strides.append(0)
            else:
                arrs.append(np.ascontiguousarray(np.stack([q['coefs'][k][..., 0] for q in ps]))); strides.append(rows)
        else:
            arrs.append(np.ascontiguousarray(np.stack([q['coefs'][k] for q in ps]))); strides.append(n)
    fl = np.tile(np.array([0., 1., 0.]), (nb, 1))
    o = _lib.options(rowconst_mask=mask)
    rc = getattr(L, util._FN[kind] + '_batched')(*[_lib.hptr(a) for a in arrs], nb, _lib.strides_arg(strides),
                                                 *util._scal(p, fl, 12, 1e-9), ctypes.byref(o))
    _lib.check(rc)
    assert np.array_equal(S, S_full) and np.array_equal(fl, fl_full)
    for m, q in enumerate(ps):
        So, flo = run_oracle(q, 12, 1e-9, COLOUR_AUTO)
        assert_same(S[m], fl[m], So, flo, 'rowconst member %d' % m)


@pytest.mark.parametrize('kind,per,bnz', [('gen2d', 'periodic', 0), ('std2d', 'fixed', 1), ('std2d', 'periodic', 0), ('bih2d', 'fixed', 1)])
def test_graph_replay_equals_plain_launches(kind, per, bnz, monkeypatch):
    """Small problems replay a captured chunk of sweep launches (hipGraph): same bits, same loop
    counts as plain launches -- colour path with the odd-width seam, fused 4-colour, fused
    red-black, biharmonic; members stopping at different sweeps inside replayed chunks."""
    xc = 251 if (kind == 'gen2d') else 144
    mk = (lambda s: randbih(40, xc, 'fixed', per, bnz, 1, seed=s)) if kind == 'bih2d' else \
         (lambda s: rand2d(kind, 73, xc, 'fixed', per, bnz, 1, seed=s))
    ps = [mk(s) for s in (1, 2, 3)]
    out = {}
    for g in ('0', '1'):                                   # xinv_options.graph: -1 = plain launches, 1 = replay
        out[g] = run_hip_batched(ps, 3000, 1e-7, check_every=16, graph=1 if g == '1' else -1)
    S0, f0, _ = out['0']; S1, f1, _ = out['1']
    assert np.array_equal(S0, S1, equal_nan=True) and np.array_equal(f0, f1, equal_nan=True)
    So, flo = run_oracle(ps[1], 3000, 1e-7, COLOUR_AUTO)
    assert np.array_equal(S1[1], So, equal_nan=True) and f1[1][2] == flo[2]


def test_dev_api_matches_host_api():
    ps = [rand2d('std2d', 48, 280, 'fixed', 'periodic', 0, 1, seed=s) for s in (1, 2, 3)]
    S1, f1, _ = run_hip_batched(ps, 40, 1e-7)
    S2, f2, st = run_hip_dev(ps, 40, 1e-7, timing=1)
    assert np.array_equal(S1, S2) and np.array_equal(f1, f2)
    assert st['sweep_ms'] > 0


def test_single_slice_positional_twins():
    for p in (rand2d('std2d', 30, 64, 'extend', 'periodic', 1, 1, seed=4),
              rand2d('gen2d', 30, 64, 'fixed', 'fixed', 0, 1, seed=5),
              rand3d(6, 12, 20, 'fixed', 'periodic', 1, seed=6)):
        So, flo = run_oracle(p, 20, 1e-9, COLOUR_AUTO)
        S, fl = run_hip_single(p, 20, 1e-9)
        assert_same(S, fl, So, flo, 'single ' + p['kind'])


def test_overflow_flag():
    """omega far outside (0,2) diverges: flags[0] set, as numbas.py:403-405."""
    p = rand2d('gen2d', 20, 40, 'fixed', 'fixed', 0, 0, seed=9, omega=40.0)
    So, flo = run_oracle(p, 2000, 1e-12, COLOUR_2)
    S, fl, _ = run_hip_batched([p], 2000, 1e-12)
    assert flo[0] == 1.0 and fl[0][0] == 1.0
    assert fl[0][2] == flo[2]


def test_restartable_in_place():
    """animate_iteration's contract (apps.py:1031-1044): two calls of n sweeps == one of 2n."""
    p = rand2d('std2d', 40, 140, 'fixed', 'periodic', 0, 1, seed=21)
    S_a, _, _ = run_hip_batched([p], 19, 0.0)          # 20 sweeps
    q = dict(p); q['S0'] = S_a[0]
    S_b, _, _ = run_hip_batched([q], 19, 0.0)          # 20 more
    S_c, _, _ = run_hip_batched([p], 39, 0.0)          # 40 at once
    assert np.array_equal(S_b, S_c)


def test_bad_arguments():
    from xinvert_amd import _lib
    p = rand2d('std2d', 2, 10, 'fixed', 'fixed', 0, 0)
    with pytest.raises(_lib.XinvError):
        run_hip_single(p, 10, 1e-9)


@pytest.mark.parametrize('path', [PATH_COLOUR, PATH_FUSED])
def test_more_members_than_one_grid_dimension(path):
    """33 000 tiny slices in one call: launches are chunked over the member axis."""
    base = rand2d('gen2d', 6, 8, 'fixed', 'periodic', 0, 0, seed=1)
    nb = 33000
    rng = np.random.default_rng(2)
    G = rng.standard_normal((nb, 6, 8))
    ps = []
    for m in range(nb):
        q = dict(base); q['coefs'] = list(base['coefs']); q['coefs'][-1] = G[m]
        ps.append(q)
    S, fl, st = run_hip_batched(ps, 6, 0.0, shared=tuple(range(6)), path=path)
    assert st['path'] == path
    for m in (0, 1, 32767, 32768, 32999):
        So, flo = run_oracle(ps[m], 6, 0.0, COLOUR_2)
        assert_same(S[m], fl[m], So, flo, 'member %d' % m)


@pytest.mark.parametrize('BCy,BCx', BCS)
@pytest.mark.parametrize('bnz', [0, 1])
@pytest.mark.parametrize('msk', [0, 1])
@pytest.mark.parametrize('shape', [(9, 12), (11, 16), (7, 7), (20, 70), (14, 17)])
def test_biharmonic_colour_path(BCy, BCx, bnz, msk, shape):
    """numbas.invert_general_bih_2D: 9 colours (+ trailing-column colours when xc % 3 != 0 and
    x is periodic), including the reference's stale-index east branches."""
    p = randbih(shape[0], shape[1], BCy, BCx, bnz, msk, seed=_seed((BCy, BCx, bnz, msk, shape)))
    So, flo = run_oracle(p, 15, 1e-9, COLOUR_AUTO)
    S, fl, st = run_hip_batched([p], 15, 1e-9, path=PATH_COLOUR)
    assert st['path'] == PATH_COLOUR
    assert st['colours'] == 9 + (3 * (shape[1] % 3) if BCx == 'periodic' else 0)
    assert_same(S[0], fl[0], So, flo, 'bih')
    # the default path: the one-pass kernel's vector-stream variants (round 6) wherever x is not periodic with xc % 3 != 0
    S1, f1 = run_hip_single(p, 15, 1e-9)
    assert np.array_equal(S1, S[0])
    Sd, fld, std = run_hip_batched([p], 15, 1e-9)
    assert std['path'] == (PATH_COLOUR if (BCx == 'periodic' and shape[1] % 3) else PATH_FUSED)
    assert np.array_equal(Sd, S) and fld[0][2] == fl[0][2]


@pytest.mark.parametrize('BCy', ['fixed', 'extend'])
@pytest.mark.parametrize('xc', [7, 8, 179, 180, 181, 185, 186, 187, 360, 366, 545])
@pytest.mark.parametrize('xuni', [0, 1])
def test_biharmonic_rowclass_strips(BCy, xc, xuni):
    """Non-periodic x runs the row-class kernel (three column colours per launch, 180-column
    strips with two halo lanes a side): strip seams, a last strip of 1..7 columns, x-uniform
    coefficient rows as per-row scalars -- all bitwise equal to the oracle's 9-colour order."""
    p = randbih(13, xc, BCy, 'fixed', 1, 1, seed=_seed((BCy, xc, xuni)))
    if xuni:
        p['coefs'] = [np.ascontiguousarray(np.broadcast_to(c[:, :1], c.shape)) if k in (0, 2, 3, 5, 8) else c
                      for k, c in enumerate(p['coefs'])]
    So, flo = run_oracle(p, 9, 1e-9, COLOUR_AUTO)
    S, fl, st = run_hip_batched([p, p], 9, 1e-9, path=PATH_COLOUR)
    assert st['path'] == PATH_COLOUR and st['colours'] == 9
    if xuni:
        assert st['xuniform_mask'] & 0x12d == 0x12d
    assert_same(S[0], fl[0], So, flo, 'bih rowclass')
    assert np.array_equal(S[0], S[1])


@pytest.mark.parametrize('BCy', ['fixed', 'extend'])
@pytest.mark.parametrize('xc', [9, 12, 177, 180, 183, 186, 360, 363, 543])
@pytest.mark.parametrize('xuni', [0, 1])
def test_biharmonic_rowclass_periodic(BCy, xc, xuni):
    """Periodic x with xc % 3 == 0 also runs the row-class kernel: wrapped lane->column map, the
    periodic branches' G-term association in the first/last two columns and the reference's
    stale-index B operand (five columns to the west) in the two east columns -- bit for bit the
    oracle's 9-colour order, across strip seams and with per-row scalar coefficients."""
    p = randbih(13, xc, BCy, 'periodic', 1, 1, seed=_seed(('bp', BCy, xc, xuni)))
    if xuni:
        p['coefs'] = [np.ascontiguousarray(np.broadcast_to(c[:, :1], c.shape)) if k in (0, 2, 3, 5, 8) else c
                      for k, c in enumerate(p['coefs'])]
    So, flo = run_oracle(p, 9, 1e-9, COLOUR_AUTO)
    S, fl, st = run_hip_batched([p, p], 9, 1e-9, path=PATH_COLOUR)
    assert st['path'] == PATH_COLOUR and st['colours'] == 9
    assert_same(S[0], fl[0], So, flo, 'bih rowclass periodic')
    assert np.array_equal(S[0], S[1])


def _uniform_bih(p):
    """A..I constant along x (Munk / Stommel-Munk coefficients are functions of latitude at most)."""
    q = dict(p)
    q['coefs'] = [np.ascontiguousarray(np.broadcast_to(c[:, :1], c.shape)) if k < 9 else c
                  for k, c in enumerate(p['coefs'])]
    return q


@pytest.mark.parametrize('BCy', ['fixed', 'extend'])
@pytest.mark.parametrize('BCx', ['fixed', 'periodic', 'extend'])
@pytest.mark.parametrize('rows', [0, 3, 9])
@pytest.mark.parametrize('shape', [(5, 9), (7, 12), (13, 183), (31, 366), (64, 543), (20, 72), (9, 180)])
@pytest.mark.parametrize('bnz', [0, 1])
def test_biharmonic_one_pass_kernel(BCy, BCx, rows, shape, bnz):
    """k_fusedbih: all nine colours of a sweep in one streaming pass (nine-row register window,
    ping-pong buffers): bit for bit the oracle's 9-colour order -- tile seams in both directions,
    first/last rows, masks, periodic wrap with the stale-index east columns, batch with early stop."""
    if BCx == 'periodic' and shape[1] % 3:
        pytest.skip('periodic x with xc % 3 != 0 runs the colour launches')
    # bnz = 0: B == E == 0 (no mixed derivatives, as in Munk): the variant that leaves those terms out
    ps = [_uniform_bih(randbih(shape[0], shape[1], BCy, BCx, bnz, 1, seed=_seed(('b1p', BCy, BCx, shape, m, bnz))))
          for m in range(2)]
    S, fl, st = run_hip_batched(ps, 25, 1e-4, rows_per_tile=rows)
    assert st['path'] == PATH_FUSED and st['colours'] == 9 and st['xuniform_mask'] & 0x1ff == 0x1ff
    for m, q in enumerate(ps):
        So, flo = run_oracle(q, 25, 1e-4, COLOUR_AUTO)
        assert_same(S[m], fl[m], So, flo, 'bih one-pass %r member %d' % (shape, m))
    Sc, flc, stc = run_hip_batched(ps, 25, 1e-4, path=PATH_COLOUR)
    assert stc['path'] == PATH_COLOUR and np.array_equal(S, Sc) and np.array_equal(fl[:, 2], flc[:, 2])


def _munk_like_bih(p):
    """A, C, D, F vary along x (A4(x, y), R(x, y): apps.py:1793-1836 puts A4 into A and C, R / D into D and F); no mixed
    derivatives (B == E == 0); G, H, I functions of the row at most."""
    q = dict(p)
    q['coefs'] = [np.zeros_like(c) if k in (1, 4) else
                  (np.ascontiguousarray(np.broadcast_to(c[:, :1], c.shape)) if k in (6, 7, 8) else c)
                  for k, c in enumerate(p['coefs'])]
    return q


@pytest.mark.parametrize('BCy', ['fixed', 'extend'])
@pytest.mark.parametrize('BCx', ['fixed', 'periodic', 'extend'])
@pytest.mark.parametrize('rows', [0, 3, 9])
@pytest.mark.parametrize('shape', [(5, 9), (7, 12), (13, 183), (31, 366), (64, 543), (20, 72), (9, 180)])
@pytest.mark.parametrize('vm', [1, 2, 3])
def test_biharmonic_one_pass_vector_streams(BCy, BCx, rows, shape, vm):
    """k_fusedbih with coefficient arrays that vary along x (round 6): vm = 1 -- A, C, D, F as vector streams beside per-row
    G, H, I (Munk with A4(x, y), R(x, y)); vm = 2 -- all nine, mixed derivatives included -- and the point-factor stream Q
    (relaxation factor, 0 = the reference's predicate on A..I forbids the update); vm = 3 -- vm = 1 with C read out of A and F
    out of D where the pairs hold the same numbers.  Bit for bit the oracle's 9-colour
    order and the colour launches: masks in coefficients and forcing, tile seams, periodic wrap with the stale-index east
    columns, a batch with an early stop."""
    if BCx == 'periodic' and shape[1] % 3:
        pytest.skip('periodic x with xc % 3 != 0 runs the colour launches')
    def alias(q):                                        # vm = 3: C holds A's numbers, F holds D's (Cartesian Munk)
        q = _munk_like_bih(q)
        q['coefs'][2] = q['coefs'][0].copy(); q['coefs'][5] = q['coefs'][3].copy()
        return q
    mk = {1: _munk_like_bih, 2: (lambda q: q), 3: alias}[vm]
    ps = [mk(randbih(shape[0], shape[1], BCy, BCx, 1, 1, seed=_seed(('bvs', BCy, BCx, shape, m, vm)))) for m in range(2)]
    S, fl, st = run_hip_batched(ps, 25, 1e-4, rows_per_tile=rows)
    assert st['path'] == PATH_FUSED and st['colours'] == 9 and st['point_factor'] == vm, st
    for m, q in enumerate(ps):
        So, flo = run_oracle(q, 25, 1e-4, COLOUR_AUTO)
        assert_same(S[m], fl[m], So, flo, 'bih vector streams vm %d %r member %d' % (vm, shape, m))
    Sc, flc, stc = run_hip_batched(ps, 25, 1e-4, path=PATH_COLOUR)
    assert stc['path'] == PATH_COLOUR and np.array_equal(S, Sc) and np.array_equal(fl[:, 2], flc[:, 2])


@pytest.mark.parametrize('BCy', ['fixed', 'extend'])
@pytest.mark.parametrize('BCx', ['fixed', 'periodic'])
@pytest.mark.parametrize('rows', [3, 9, 12])
def test_biharmonic_one_pass_skips_masked_tiles(BCy, BCx, rows):
    """Masked-tile skipping in k_fusedbih (land in an ocean Munk problem): tiles whose forcing is
    undefined throughout are left out, their norm share is constant -- bit for bit the oracle and the
    run that visits every tile; a last row block of a single row with 'extend' (rows yc-2, yc-1 sit
    in two different blocks)."""
    rng = np.random.default_rng(_seed(('bskip', BCy, BCx, rows)))
    yc, xc = 3 * 13 + 1, 3 * 170
    ps = []
    for m in range(2):
        q = _uniform_bih(randbih(yc, xc, BCy, BCx, 0, 1, seed=_seed(('bskip', m))))
        J = q['coefs'][9]
        J[: yc // 2, : xc // 2] = util.U
        if m == 1:
            J[yc // 2:, 200:] = util.U
        q['S0'][: yc // 2, : xc // 2] = np.where(rng.random((yc // 2, xc // 2)) < 0.2, util.U, 0.5)
        ps.append(q)
    S, fl, st = run_hip_batched(ps, 30, 1e-5, rows_per_tile=rows, force_tile_skip=1)
    # ('extend' keeps the first and the last two row blocks active: with 12-row blocks nothing is left to skip)
    assert st['path'] == PATH_FUSED and (st['masked_tile_pct'] > 0 or (BCy == 'extend' and rows == 12))
    S0, fl0, st0 = run_hip_batched(ps, 30, 1e-5, rows_per_tile=rows, no_tile_skip=1)
    assert st0['masked_tile_pct'] == 0 and np.array_equal(S, S0) and np.array_equal(fl[:, 2], fl0[:, 2])
    for m, q in enumerate(ps):
        So, flo = run_oracle(q, 30, 1e-5, COLOUR_AUTO)
        assert_same(S[m], fl[m], So, flo, 'bih skip member %d' % m)


def test_biharmonic_batched_dev():
    ps = [randbih(16, 33, 'extend', 'periodic', 1, 1, seed=s) for s in (3, 4)]
    S1, f1, _ = run_hip_batched(ps, 20, 1e-7)
    S2, f2, _ = run_hip_dev(ps, 20, 1e-7)
    assert np.array_equal(S1, S2) and np.array_equal(f1, f2)
    for m, q in enumerate(ps):
        So, flo = run_oracle(q, 20, 1e-7, COLOUR_AUTO)
        assert_same(S1[m], f1[m], So, flo, 'bih member %d' % m)


@pytest.mark.parametrize('kind', ['std2d', 'gen2d'])
@pytest.mark.parametrize('path', [PATH_COLOUR, PATH_FUSED])
def test_degenerate_inputs_follow_the_reference_loop_control(kind, path):
    """(i) every point masked: S stays 0, norm == 0 -> standard_2D stops at once (numbas.py:410),
    general_2D runs to mxLoop with flags[1] = |0-0|/0 = NaN; (ii) S entirely undef: the norm has
    and nothing updatable: the norm has no sample -> NaN -> overflow exit on the first sweep
    (numbas.py:403-405, 1723-1726);
    (iii) a NaN coefficient poisons S and trips the same exit."""
    p = rand2d(kind, 20, 36, 'fixed', 'periodic', 0, 0, seed=3)
    q = dict(p); q['coefs'] = list(p['coefs']); q['coefs'][-1] = np.full((20, 36), util.U)
    q['S0'] = np.zeros((20, 36))
    So, flo = run_oracle(q, 7, 1e-9, COLOUR_2)
    S, fl, _ = run_hip_batched([q], 7, 1e-9, path=path)
    assert_same(S[0], fl[0], So, flo, 'all masked')
    assert flo[2] == (0 if kind == 'std2d' else 7)
    r = dict(q); r['S0'] = np.full((20, 36), util.U)      # nothing defined, nothing updated
    So, flo = run_oracle(r, 7, 1e-9, COLOUR_2)
    S, fl, _ = run_hip_batched([r], 7, 1e-9, path=path)
    assert flo[0] == 1.0 and fl[0][0] == 1.0 and fl[0][2] == flo[2] == 0.0
    r2 = dict(p); r2['S0'] = np.full((20, 36), util.U)    # S undef but points updatable: S is
    So, flo = run_oracle(r2, 7, 1e-9, COLOUR_2)            # never tested by the update predicate
    S, fl, _ = run_hip_batched([r2], 7, 1e-9, path=path)
    assert_same(S[0], fl[0], So, flo, 'S all undef')
    t = dict(p); t['coefs'] = [c.copy() for c in p['coefs']]; t['coefs'][0][5, 7] = np.nan
    So, flo = run_oracle(t, 50, 1e-9, COLOUR_2)
    S, fl, _ = run_hip_batched([t], 50, 1e-9, path=path)
    assert flo[0] == 1.0 and fl[0][0] == 1.0 and fl[0][2] == flo[2]
    # same exit, same sweep; the NaN footprint may differ: with B == 0 the engine skips the
    # reference's cross terms 0 * (S - S), which propagate NaN diagonally only once S is poisoned
    assert np.isnan(S[0][5, 7]) and np.isnan(So[5, 7])


def test_mxloop_zero_does_one_sweep():
    p = rand2d('gen2d', 12, 20, 'extend', 'fixed', 0, 1, seed=8)
    So, flo = run_oracle(p, 0, 1e-9, COLOUR_2)
    S, fl, st = run_hip_batched([p], 0, 1e-9)
    assert_same(S[0], fl[0], So, flo, 'mxLoop=0')
    assert st['sweeps_max'] == 1


@pytest.mark.parametrize('BCy,BCx', BCS)
@pytest.mark.parametrize('bnz', [0, 1])
@pytest.mark.parametrize('msk', [0, 1])
@pytest.mark.parametrize('shape', [(9, 12), (12, 19), (30, 260)])
def test_standard_2d_test_form(BCy, BCx, bnz, msk, shape):
    """numbas.invert_standard_2D_test: colour path (5/9-point, seam) and, when B == C == 0,
    the fused kernels (K = 1, 2, 3; full-array and x-uniform variants)."""
    p = rand2dt(shape[0], shape[1], BCy, BCx, bnz, msk, seed=_seed((BCy, BCx, bnz, msk, shape)))
    So, flo = run_oracle(p, 20, 1e-9, COLOUR_AUTO)
    S, fl, st = run_hip_batched([p], 20, 1e-9, path=PATH_COLOUR)
    assert_same(S[0], fl[0], So, flo, 'std2dt colour')
    if bnz == 0 and not (BCx == 'periodic' and shape[1] % 2):
        for K in (1, 2, 3):
            S, fl, st = run_hip_batched([p], 20, 1e-9, path=PATH_FUSED, sweeps_per_launch=K, rows_per_tile=10)
            assert st['path'] == PATH_FUSED and st['sweeps_per_launch'] == K
            assert_same(S[0], fl[0], So, flo, 'std2dt fused K=%d' % K)
        if not msk:
            q = dict(p)
            q['coefs'] = [np.ascontiguousarray(np.broadcast_to(c[:, :1], c.shape)) if k in (0, 3, 4) else c
                          for k, c in enumerate(p['coefs'])]
            So, flo = run_oracle(q, 20, 1e-9, COLOUR_AUTO)
            for K in (0, 1, 2, 3):
                S, fl, st = run_hip_batched([q], 20, 1e-9, sweeps_per_launch=K)
                assert st['xuniform_mask'] == 7 and (K == 0 or st['sweeps_per_launch'] == K)
                assert_same(S[0], fl[0], So, flo, 'std2dt fused x-uniform K=%d' % K)


def test_concurrent_host_threads_are_serialised_safely():
    """Two host threads solving on the same device share the cached workspace: the per-device
    lock must keep both results exact."""
    import threading
    ps = [rand2d('gen2d', 60, 300, 'fixed', 'periodic', 0, 1, seed=s) for s in (31, 32)]
    want = [run_oracle(p, 60, 0.0, COLOUR_2)[0] for p in ps]
    got = [None, None]

    def work(k):
        for _ in range(5):
            got[k] = run_hip_batched([ps[k]], 60, 0.0)[0][0]

    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    [t.start() for t in th]; [t.join() for t in th]
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])


NINE_SHAPES = [(17, 24), (12, 20), (40, 300), (33, 257), (9, 8), (3, 3), (50, 512), (21, 130)]


@pytest.mark.parametrize('kind,K', [('std2d', 1), ('std2d', 2), ('std2d', 3), ('gen2d', 1), ('gen2d', 2)])
@pytest.mark.parametrize('BCy,BCx', BCS)
@pytest.mark.parametrize('msk', [0, 1])
@pytest.mark.parametrize('shape', NINE_SHAPES)
def test_fused_nine_point(kind, K, BCy, BCx, msk, shape):
    """B != 0: the fused 4-colour kernel against the oracle's 4-colour ordering (incl. the
    reference's west-periodic operands of invert_standard_2D)."""
    yc, xc = shape
    if BCx == 'periodic' and xc % 2:
        pytest.skip('odd-xc periodic seam goes through the colour path')
    p = rand2d(kind, yc, xc, BCy, BCx, 1, msk, seed=_seed((kind, BCy, BCx, msk, shape, 9)), omega=1.1)
    So, flo = run_oracle(p, 17, 1e-9, COLOUR_AUTO)
    S, fl, st = run_hip_batched([p], 17, 1e-9, path=PATH_FUSED, sweeps_per_launch=K, rows_per_tile=10)
    assert st['path'] == PATH_FUSED and st['colours'] == 4 and st['sweeps_per_launch'] == K
    assert_same(S[0], fl[0], So, flo, 'fused 9-point K=%d %s %r' % (K, kind, shape))


def test_fused_nine_point_default_tiling_and_batch():
    ps = [rand2d('gen2d', 70, 380, 'extend', 'periodic', 1, 1, seed=s, omega=1.1) for s in (41, 42, 43)]
    S, fl, st = run_hip_batched(ps, 200, 3e-4)
    assert st['path'] == PATH_FUSED and st['colours'] == 4
    S2, fl2, st2 = run_hip_batched(ps, 200, 3e-4, path=PATH_COLOUR)
    assert np.array_equal(S, S2) and np.allclose(fl, fl2, rtol=1e-9, atol=1e-12)
    for m, q in enumerate(ps):
        So, flo = run_oracle(q, 200, 3e-4, COLOUR_AUTO)
        assert_same(S[m], fl[m], So, flo, '9-point member %d' % m)


@pytest.mark.parametrize('chunk', range(32))
def test_seeded_fuzz_against_the_oracle(chunk):
    """Seeded random configurations (kind x shape x BCs x mask density x B != 0 x engine options) --
    whatever path the engine picks must reproduce the oracle's coloured ordering bit for bit."""
    rng = np.random.default_rng(9000 + chunk)
    for case in range(24):
        kind = ['std2d', 'gen2d', 'std2dt', 'bih2d', 'std3d', 'gen3d'][int(rng.integers(6))]
        BCy = ['fixed', 'extend'][int(rng.integers(2))]
        BCx = ['fixed', 'periodic', 'extend'][int(rng.integers(3))]
        msk = int(rng.integers(2)); bnz = int(rng.integers(2))
        seed = int(rng.integers(1 << 30))
        if kind in ('std3d', 'gen3d'):
            zc, yc, xc = int(rng.integers(3, 9)), int(rng.integers(3, 40)), int(rng.integers(3, 300))
            p = (rand3d if kind == 'std3d' else rand3dg)(zc, yc, xc, BCy, BCx, msk, seed=seed)
        elif kind == 'bih2d':
            yc, xc = int(rng.integers(5, 60)), int(rng.integers(7, 400))
            p = randbih(yc, xc, BCy, BCx, bnz, msk, seed=seed)
        else:
            yc, xc = int(rng.integers(3, 90)), int(rng.integers(3, 520))
            p = rand2dt(yc, xc, BCy, BCx, bnz, msk, seed=seed) if kind == 'std2dt' else \
                rand2d(kind, yc, xc, BCy, BCx, bnz, msk, seed=seed)
        if msk and kind in ('std2d', 'gen2d', 'std2dt') and int(rng.integers(2)):
            j0 = int(rng.integers(0, max(1, yc // 2))); i0 = int(rng.integers(0, max(1, xc // 2)))
            p = _blocky(p, rng, [(j0, yc, i0, xc)])
        if int(rng.integers(2)):                   # coefficients constant along x: the streaming variants
            if kind == 'bih2d':
                p = _uniform_bih(p)
            elif kind == 'gen3d':
                p = _uniform3dg(p)
            elif kind == 'std3d':
                p = _uniform3d(p, None)
            elif kind == 'std2d' and not bnz:
                p = _uniform2d(p)
        opt = {}
        if kind in ('bih2d', 'std3d', 'gen3d') and int(rng.integers(2)):
            opt['rows_per_tile'] = [3, 9, 27][int(rng.integers(3))] if kind == 'bih2d' else [8, 12, 16][int(rng.integers(3))]
        if kind in ('std2d', 'gen2d', 'std2dt'):
            opt['sweeps_per_launch'] = int(rng.integers(0, 5))     # capped at what the variant supports
            if yc >= 8 and int(rng.integers(2)):
                opt['rows_per_tile'] = -int(rng.integers(1, max(2, yc // 4) + 1))
            opt['force_tile_skip'] = int(rng.integers(2))
            opt['no_xuniform'] = int(rng.integers(2))
        nsw = int(rng.integers(1, 14)); tol = [0.0, 1e-3][int(rng.integers(2))]
        So, flo = run_oracle(p, nsw, tol, COLOUR_AUTO)
        try:
            S, fl, st = run_hip_batched([p], nsw, tol, **opt)
        except Exception as e:                       # general 9-point form has no K = 2 kernel
            if 'sweeps_per_launch' in str(e) or 'unsupported' in str(e):
                opt.pop('sweeps_per_launch', None)
                S, fl, st = run_hip_batched([p], nsw, tol, **opt)
            else:
                raise
        what = 'fuzz %d/%d %s %r %s %s msk=%d bnz=%d %r' % (chunk, case, kind, p['S0'].shape, BCy, BCx, msk, bnz, opt)
        if np.isnan(So).any():
            assert np.array_equal(S[0], So, equal_nan=True), what
        else:
            assert_same(S[0], fl[0], So, flo, what)


@pytest.mark.parametrize('BCx', ['fixed', 'periodic'])
@pytest.mark.parametrize('shape', [(123, 813), (242, 219), (60, 399)])
def test_bih_extend_tolerance_stop_with_the_lagged_norm(shape, BCx):
    """The one-pass biharmonic kernel's 'extend' pre-pass works in place on its launch's source buffer; with the lagged
    norm the launch AFTER the one the stop rule fires in has already run and must not leave its pre-pass in the final
    state (rows 0, 1, yc-2, yc-1; found by the extended medium fuzz, chunks 114 ... 248, at the end of round 4)."""
    yc, xc = shape
    ps = [_uniform_bih(randbih(yc, xc, 'extend', BCx, 0, 1, seed=_seed(('bihlag', shape, BCx, m)))) for m in range(2)]
    stopped = []
    for tol in (3e-3, 1e-2, 3e-2):
        S, fl, st = run_hip_batched(ps, 60, tol)
        assert st['path'] == PATH_FUSED, st
        for m, q in enumerate(ps):
            So, flo = run_oracle(q, 60, tol, COLOUR_AUTO)
            assert_same(S[m], fl[m], So, flo, 'bih extend tol=%g %r member %d' % (tol, shape, m))
            stopped.append(int(flo[2]))
    assert min(stopped) < 60, stopped                    # (some case really stopped on the tolerance)


@pytest.mark.parametrize('chunk', range(16))
def test_seeded_fuzz_medium_grids(chunk):
    """Larger seeded cases (hundreds of rows, several strips and row blocks, batches of two with a
    shared coefficient stack): automatic tiling, masked-tile skipping, k chunks and the x-uniform
    variants all engage on their own here."""
    _medium_fuzz(chunk, 7000, 0)


@pytest.mark.parametrize('chunk', range(12))
def test_seeded_fuzz_medium_grids_odd_widths(chunk):
    """The same generator with ODD widths (the biharmonic form keeps multiples of three): with periodic x the seam
    variants of every streaming kernel -- 5-point, 9-point, both 3-D forms -- chosen by the engine itself."""
    _medium_fuzz(chunk, 9000, 1)


def _medium_fuzz(chunk, seed0, odd, every_case_stops=False, runner=None):
    rng = np.random.default_rng(seed0 + chunk)
    for case in range(4):
        kind = ['std2d', 'gen2d', 'std2dt', 'bih2d', 'std3d', 'gen3d'][int(rng.integers(6))]
        BCy = ['fixed', 'extend'][int(rng.integers(2))]
        BCx = ['fixed', 'periodic'][int(rng.integers(2))]
        seed = int(rng.integers(1 << 30))
        uni = int(rng.integers(2))
        if kind in ('std3d', 'gen3d'):
            zc, yc, xc = int(rng.integers(20, 60)), int(rng.integers(30, 80)), 2 * int(rng.integers(60, 200)) + odd
            mk = (lambda s: rand3d(zc, yc, xc, BCy, BCx, 1, seed=s)) if kind == 'std3d' else \
                 (lambda s: rand3dg(zc, yc, xc, BCy, BCx, 1, seed=s))
            un = (lambda q: _uniform3d(q, None)) if kind == 'std3d' else _uniform3dg
        elif kind == 'bih2d':
            yc, xc = int(rng.integers(60, 300)), 3 * int(rng.integers(60, 400))
            mk = lambda s: randbih(yc, xc, BCy, BCx, int(rng.integers(2)), 1, seed=s)
            un = _uniform_bih
        else:
            yc, xc = int(rng.integers(100, 400)), 2 * int(rng.integers(100, 700)) + odd
            bnz = int(rng.integers(2))                 # 1: cross terms -> 4-colour kernels
            mk = (lambda s: rand2dt(yc, xc, BCy, BCx, bnz, 1, seed=s)) if kind == 'std2dt' else \
                 (lambda s: rand2d(kind, yc, xc, BCy, BCx, bnz, 1, seed=s))
            un = None
        ps = [mk(seed), mk(seed + 1)]
        if kind in ('std2d', 'gen2d', 'std2dt'):
            ps = [_blocky(q, rng, [(0, yc // 2, 0, xc // 3), (yc // 2, yc, xc // 2, xc)]) for q in ps]
            if uni:                                   # latitude-only coefficients
                ps = [dict(q, coefs=[np.ascontiguousarray(np.broadcast_to(c[:, :1], c.shape))
                                     if k < len(q['coefs']) - 1 else c for k, c in enumerate(q['coefs'])])
                      for q in ps]
        elif uni:
            ps = [un(q) for q in ps]
            if kind == 'bih2d':                       # land blocks in the forcing: skippable tiles
                ps = [_blocky(q, rng, [(0, yc // 2, 0, xc // 3), (yc // 2, yc, xc // 2, xc)]) for q in ps]
        for k in range(len(ps[0]['coefs']) - 1):      # one coefficient stack for both members
            ps[1]['coefs'][k] = ps[0]['coefs'][k]
        shared = tuple(range(len(ps[0]['coefs']) - 1))
        nsw = int(rng.integers(2, 8))
        tol = 0.0
        if case == 3 or every_case_stops:             # one case per chunk stops on the tolerance (odd sweep
            nsw, tol = 60, 3e-3                       # counts inside 2-sweep launches: the redo path)
        opt = {'force_tile_skip': 1} if kind in ('std2d', 'gen2d', 'std2dt', 'bih2d') and int(rng.integers(2)) else {}
        S, fl, st = (runner or run_hip_batched)(ps, nsw, tol, shared=shared, **opt)      # (runner: tests/fuzz_plan.py)
        for m, q in enumerate(ps):
            So, flo = run_oracle(q, nsw, tol, COLOUR_AUTO)
            what = 'medium fuzz %d/%d %s %r %s %s uni=%d member %d %r' % (chunk, case, kind, q['S0'].shape, BCy, BCx, uni, m, st)
            if np.isnan(So).any():
                assert np.array_equal(S[m], So, equal_nan=True), what
            else:
                assert_same(S[m], fl[m], So, flo, what)


@pytest.mark.parametrize('BCy', ['fixed', 'extend'])
@pytest.mark.parametrize('BCx', ['fixed', 'periodic'])
@pytest.mark.parametrize('nmem,cus', [(3, 8), (7, 8), (2, 5), (4, 256)])
@pytest.mark.parametrize('shape', [(50, 40, 130), (37, 20, 250), (50, 41, 130), (50, 42, 130)])
def test_pipe3d_tail_cut(BCy, BCx, nmem, cus, shape):
    """k_pipe3d runs one workgroup per CU; the tiles of a launch's last, partly filled round are cut into k chunks while the
    others march the whole column (xinv_pipe3d.h; xinv_tiles.h: xinv_p3_whole_tiles).  With `cu_count` a small batch takes the mixed launch:
    whole-column and cut tiles of one member, members of both kinds -- bit for bit the oracle, stop rule included."""
    # (BCy = 'extend': yc = 40 / 42 put rows yc-2, yc-1 into one wavefront; 41 splits them: the one-sweep kernel)
    ps = [_uniform3d(rand3d(shape[0], shape[1], shape[2], BCy, BCx, 1, seed=_seed(('tail', BCy, BCx, nmem, cus, shape, m))), None)
          for m in range(nmem)]
    S, fl, st = run_hip_batched(ps, 40, 1e-4, path=PATH_FUSED, cu_count=cus, lanes=1)
    two = BCy == 'fixed' or util.p3_extend_ok(shape[1])
    assert st['path'] == PATH_FUSED and st['sweeps_per_launch'] == (2 if two else 1), st
    if cus < 256 and two:
        assert st['k_chunks'] > 1 and 0 < st['cut_tiles'] < cus, st
    for m, q in enumerate(ps):
        So, flo = run_oracle(q, 40, 1e-4, COLOUR_2)
        assert_same(S[m], fl[m], So, flo, 'tail cut %r member %d' % (shape, m))
