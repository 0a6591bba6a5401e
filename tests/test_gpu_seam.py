"""Periodic x with ODD xc (the reference's own tests/test_Ishida.py: xnum = 251) on the streaming kernels.

Columns 0 and xc-1 are neighbours of one colour there.  The coloured ordering (oracle: seq_colour) updates column
xc-1 inside the half-sweep of its own colour, right after column 0 -- red, red', black, black' -- and the fused
kernels do the same on the even-ring layout: a phantom column mirroring column xc-1, one more pass for the seam lanes in
the tiles that hold them (xinv_fused.h: RING / SEAM; halos: xinv_tiles.h).  Bit for bit
against the oracle and the colour launches, single-strip rows (65 ... 127 columns: the strip wraps on both sides),
multi-strip rows, every sweeps-per-pass, masks, 'extend', x-uniform and full coefficient arrays, batches."""
import zlib

import numpy as np
import pytest

from util import rand2d, rand2dt, rand3d, rand3dg, run_oracle, run_hip_batched

pytestmark = pytest.mark.gpu
COLOUR_2, PATH_COLOUR, PATH_FUSED = 2, 1, 2
# (113 .. 125: either side of the widths from which the ring layout's strips take the asymmetric halos, 126 - H for H = 2 ..
#  12 halo columns -- xinv_tiles.h; 219, 221, 235: a last strip of one or two columns beside them)
SHAPES = [(20, 65), (70, 101), (30, 127), (25, 129), (40, 301), (33, 257), (12, 641), (22, 113), (22, 115), (22, 117), (22, 119),
          (22, 121), (22, 123), (22, 125), (26, 219), (26, 235)]
# widths that put a FULL strip next to the seam for one of the tilings (seam strips own 128 - 4K - 2 columns: 122, 118,
# 114, 110): column xc-1 is updated after column 0 inside one half-sweep, so the dependency cone of the columns west of
# the seam reaches 2K + 1 columns east across it -- one more than the plain halo (found in round 4 with 361 = 3 x 120 + 1)
EDGE_WIDTHS = [(24, 245), (24, 237), (24, 229), (24, 221), (24, 331), (24, 343), (24, 361), (24, 363), (24, 359), (24, 241)]


def _seed(t):
    return zlib.crc32(repr(t).encode()) % 100000


def _same(S, fl, So, flo, what):
    assert np.array_equal(S, So), '%s: %d points differ' % (what, (S != So).sum())
    assert fl[2] == flo[2] and fl[0] == flo[0] and abs(fl[1] - flo[1]) <= 1e-12 * max(1.0, abs(flo[1])), (what, fl, flo)


def _uniform(p, which):
    q = dict(p)
    q['coefs'] = [np.ascontiguousarray(np.broadcast_to(c[:, :1], c.shape)) if k in which else c
                  for k, c in enumerate(p['coefs'])]
    return q


@pytest.mark.parametrize('kind', ['std2d', 'gen2d'])
@pytest.mark.parametrize('BCy', ['fixed', 'extend'])
@pytest.mark.parametrize('msk', [0, 1])
@pytest.mark.parametrize('shape', SHAPES + EDGE_WIDTHS)
def test_seam_fused_full_arrays(kind, BCy, msk, shape):
    yc, xc = shape
    p = rand2d(kind, yc, xc, BCy, 'periodic', 0, msk, seed=_seed((kind, BCy, msk, shape)))
    So, flo = run_oracle(p, 13, 1e-9, COLOUR_2)
    Sc, fc, sc = run_hip_batched([p], 13, 1e-9, path=PATH_COLOUR)
    assert sc['path'] == PATH_COLOUR and sc['colours'] == 4
    _same(Sc[0], fc[0], So, flo, 'colour launches')
    for K in (1, 2, 3, 4) if kind == 'std2d' else (1, 2, 3):
        for rows in (16, 0):
            S, fl, st = run_hip_batched([p], 13, 1e-9, path=PATH_FUSED, sweeps_per_launch=K, rows_per_tile=rows)
            assert st['path'] == PATH_FUSED and st['sweeps_per_launch'] == K, st
            _same(S[0], fl[0], So, flo, 'fused K=%d rows=%d %s %r' % (K, rows, kind, shape))


@pytest.mark.parametrize('kind', ['std2d', 'gen2d'])
@pytest.mark.parametrize('BCy', ['fixed', 'extend'])
@pytest.mark.parametrize('shape', SHAPES + [(151, 251)] + EDGE_WIDTHS)
def test_seam_fused_x_uniform_and_batches(kind, BCy, shape):
    """Per-row coefficients (lat-lon Poisson / Gill-Matsuno; the Ishida case is a general form with constants): the
    engine's own choice of kernel, three members with different masks, tolerance stops inside a pass."""
    yc, xc = shape
    which = (0, 2) if kind == 'std2d' else (0, 2, 3, 4, 5)
    ps = [_uniform(rand2d(kind, yc, xc, BCy, 'periodic', 0, m & 1, seed=_seed((kind, BCy, shape, m))), which) for m in range(3)]
    ref = [run_oracle(p, 200, 2e-4, COLOUR_2) for p in ps]
    for kw in (dict(), dict(no_pipe=1), dict(sweeps_per_launch=2), dict(force_tile_skip=1)):
        S, fl, st = run_hip_batched(ps, 200, 2e-4, **kw)
        assert st['path'] == PATH_FUSED and st['xuniform_mask'] == (3 if kind == 'std2d' else 31), st
        for m in range(3):
            _same(S[m], fl[m], ref[m][0], ref[m][1], '%s %r member %d %r' % (kind, shape, m, kw))


@pytest.mark.parametrize('BCy', ['fixed', 'extend'])
@pytest.mark.parametrize('shape', [(30, 261), (20, 77)])
def test_seam_fused_test_form(BCy, shape):
    p = rand2dt(shape[0], shape[1], BCy, 'periodic', 0, 1, seed=_seed((BCy, shape)))
    So, flo = run_oracle(p, 20, 1e-9, COLOUR_2)
    for K in (1, 2, 3):
        S, fl, st = run_hip_batched([p], 20, 1e-9, path=PATH_FUSED, sweeps_per_launch=K, rows_per_tile=10)
        assert st['path'] == PATH_FUSED and st['sweeps_per_launch'] == K
        _same(S[0], fl[0], So, flo, 'std2dt K=%d' % K)


@pytest.mark.parametrize('kind', ['std2d', 'gen2d'])
@pytest.mark.parametrize('shape', [(300, 257), (180, 361), (402, 113), (96, 1201)])
def test_seam_edge_strips_in_half_height_tiles(kind, shape):
    """Tall row blocks, the edge strips' tiles dispatched first (round 4 cut their row blocks in pieces; the ring layout's
    launches keep them whole: profiles/r05_seam_rates.txt) -- the dispatch order in the launches and in the masked-tile
    lists, the skipped tiles' norm share; fixed and even row splits, one strip spanning the row (113)
    and many."""
    yc, xc = shape
    which = (0, 2) if kind == 'std2d' else (0, 2, 3, 4, 5)
    ps = [_uniform(rand2d(kind, yc, xc, 'fixed', 'periodic', 0, 1, seed=_seed((kind, shape, m))), which) for m in range(2)]
    for q in ps:                                       # blank blocks of the forcing: whole tiles (and halves) to skip
        F = q['coefs'][-1]
        F[yc // 3:yc // 3 + yc // 4, :xc // 2] = q['undef']
        F[:, xc - 40:xc - 8][yc // 2:] = q['undef']
    ref = [run_oracle(p, 29, 1e-9, COLOUR_2) for p in ps]
    for kw in (dict(), dict(force_tile_skip=1), dict(rows_per_tile=64), dict(rows_per_tile=-3, force_tile_skip=1),
               dict(no_pipe=1, force_tile_skip=1), dict(sweeps_per_launch=3)):
        S, fl, st = run_hip_batched(ps, 29, 1e-9, **kw)
        assert st['path'] == PATH_FUSED, st
        for m in range(2):
            _same(S[m], fl[m], ref[m][0], ref[m][1], '%s %r member %d %r' % (kind, shape, m, kw))


@pytest.mark.parametrize('kind', ['std2d', 'gen2d'])
@pytest.mark.parametrize('BCy', ['fixed', 'extend'])
@pytest.mark.parametrize('msk', [0, 1])
@pytest.mark.parametrize('shape', SHAPES + [(24, 111), (24, 221), (24, 331), (24, 95), (24, 189), (24, 283), (151, 251)])
def test_seam_fused_nine_point(kind, BCy, msk, shape):
    """B != 0: the 4-colour kernel with the seam colours c0' / c2' (k_fused9's SEAM variants: lanes classed by the wrapped
    column of each slot, up to six masked passes per row stage, 128 - 8K - 2 owned columns: 118, 110, 102 -- the widths
    111, 221, 331 / 95, 189, 283 put a full strip next to the seam); the standard form's i == 0 branch (numbas.py:327-328)
    on a column 0 that sits in a lane's .y slot."""
    yc, xc = shape
    ps = [rand2d(kind, yc, xc, BCy, 'periodic', 1, (msk + m) & 1, seed=_seed(('nine', kind, BCy, msk, shape, m))) for m in range(2)]
    ref = [run_oracle(p, 13, 1e-9, 4) for p in ps]
    Sc, fc, sc = run_hip_batched(ps, 13, 1e-9, path=PATH_COLOUR)
    assert sc['path'] == PATH_COLOUR and sc['colours'] == 6
    for m in range(2):
        _same(Sc[m], fc[m], ref[m][0], ref[m][1], 'colour launches')
    for K in (1, 2, 3) if kind == 'std2d' else (1, 2):
        for rows in (16, 0):
            S, fl, st = run_hip_batched(ps, 13, 1e-9, path=PATH_FUSED, sweeps_per_launch=K, rows_per_tile=rows)
            assert st['path'] == PATH_FUSED and st['sweeps_per_launch'] == K and st['colours'] == 6, st
            for m in range(2):
                _same(S[m], fl[m], ref[m][0], ref[m][1], 'nine-point K=%d rows=%d %s %r member %d' % (K, rows, kind, shape, m))


# 3-D standard form (k_fused3d's and k_pipe3d's SEAM variants on the ring layout: 120 / 122 and 116 / 118 owned columns).  Widths: one strip wrapping on both sides, a full strip next to the seam (245 = 2 x 122 + 1, 123, 367),
# many strips; heights around the 8 / 4 owned rows of the 12- / 8-wavefront cross-sections; k chunks (tall volumes).
SHAPES_3D = [(7, 20, 65), (9, 23, 101), (6, 17, 127), (12, 30, 129), (8, 19, 245), (8, 14, 123), (5, 9, 367), (40, 11, 131),
             (6, 12, 641), (6, 15, 119), (6, 15, 121), (6, 15, 125), (6, 15, 237)]


@pytest.mark.parametrize('BCy', ['fixed', 'extend'])
@pytest.mark.parametrize('msk', [0, 1])
@pytest.mark.parametrize('shape', SHAPES_3D)
def test_seam_fused_3d(BCy, msk, shape):
    zc, yc, xc = shape
    p = rand3d(zc, yc, xc, BCy, 'periodic', msk, seed=_seed((BCy, msk, shape)))
    So, flo = run_oracle(p, 11, 1e-9, COLOUR_2)
    Sc, fc, sc = run_hip_batched([p], 11, 1e-9, path=PATH_COLOUR)
    assert sc['path'] == PATH_COLOUR and sc['colours'] == 4
    _same(Sc[0], fc[0], So, flo, 'colour launches')
    for rows in (0, 8, 12):                             # full coefficient arrays
        S, fl, st = run_hip_batched([p], 11, 1e-9, path=PATH_FUSED, rows_per_tile=rows)
        assert st['path'] == PATH_FUSED and st['sweeps_per_launch'] == 1 and st['xuniform_mask'] == 0, st
        _same(S[0], fl[0], So, flo, '3-D rows=%d %r' % (rows, shape))
    # x-uniform coefficients (every lat-lon omega problem), three members, the engine's own choice, a tolerance stop
    qs = []
    for m in range(3):
        q = rand3d(zc, yc, xc, BCy, 'periodic', (msk + m) & 1, seed=_seed((BCy, msk, shape, m)))
        q['coefs'] = [np.ascontiguousarray(np.broadcast_to(c[:, :, :1], c.shape)) if k < 3 else c
                      for k, c in enumerate(q['coefs'])]
        qs.append(q)
    # (round 5: with BCy = 'fixed' the engine's own choice is the two-sweep pass, k_pipe3d's ring variant -- the row as an
    #  even ring with a phantom column; sweeps_per_launch = 1 / a forced cross-section keep k_fused3d's seam variants;
    #  61 sweeps: an odd count ends with one pass of the one-sweep kernel behind the two-sweep passes)
    two = 2 if BCy == 'fixed' else 1
    for mx, tol in ((60, 1e-3), (61, 0.0)):
        ref = [run_oracle(q, mx, tol, COLOUR_2) for q in qs]
        for kw, K in ((dict(), two), (dict(rows_per_tile=8), 1), (dict(sweeps_per_launch=2), two), (dict(sweeps_per_launch=1), 1)):
            S, fl, st = run_hip_batched(qs, mx, tol, **kw)
            assert st['path'] == PATH_FUSED and st['xuniform_mask'] == 7 and st['sweeps_per_launch'] == K, st
            for m in range(3):
                _same(S[m], fl[m], ref[m][0], ref[m][1], '3-D x-uniform %r member %d %r mxLoop %d' % (shape, m, kw, mx))


@pytest.mark.parametrize('BCy', ['fixed', 'extend'])
@pytest.mark.parametrize('msk', [0, 1])
@pytest.mark.parametrize('shape', SHAPES_3D)
def test_seam_fused_3d_general_form(BCy, msk, shape):
    """invert_general_3D with every coefficient constant along x (invert_3DOcean): k_fused3dg's SEAM variants, incl. the
    reference's i == 0 branch that never tests the forcing (numbas.py:849-852) on the wrapped copies of column 0."""
    zc, yc, xc = shape
    qs = []
    for m in range(2):
        q = rand3dg(zc, yc, xc, BCy, 'periodic', (msk + m) & 1, seed=_seed(('g', BCy, msk, shape, m)))
        q['coefs'] = [np.ascontiguousarray(np.broadcast_to(c[:, :, :1], c.shape)) if k < 7 else c
                      for k, c in enumerate(q['coefs'])]
        qs.append(q)
    ref = [run_oracle(q, 14, 1e-9, COLOUR_2) for q in qs]
    Sc, fc, sc = run_hip_batched(qs, 14, 1e-9, path=PATH_COLOUR)
    assert sc['path'] == PATH_COLOUR and sc['colours'] == 4
    for kw in (dict(), dict(rows_per_tile=8)):
        S, fl, st = run_hip_batched(qs, 14, 1e-9, **kw)
        assert st['path'] == PATH_FUSED and st['xuniform_mask'] == 127, st
        for m in range(2):
            _same(Sc[m], fc[m], ref[m][0], ref[m][1], 'colour launches')
            _same(S[m], fl[m], ref[m][0], ref[m][1], 'general 3-D %r member %d %r' % (shape, m, kw))


def test_short_odd_rows_keep_the_colour_launches():
    p = rand2d('std2d', 20, 33, 'fixed', 'periodic', 0, 1, seed=5)
    So, flo = run_oracle(p, 30, 1e-9, COLOUR_2)
    S, fl, st = run_hip_batched([p], 30, 1e-9)
    assert st['path'] == PATH_COLOUR
    _same(S[0], fl[0], So, flo, 'xc = 33')
    with pytest.raises(Exception, match='no fused kernel'):
        run_hip_batched([p], 3, 0.0, path=PATH_FUSED)
