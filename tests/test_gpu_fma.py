"""XINV_FLAG_FMA (opt-in contracted arithmetic): the HIP kernels with the flag against the oracle's XO_FMA restatement of
the same ordering, BIT FOR BIT -- pipelined passes, tail passes, tolerance stops inside a pass, masks, 'extend', batches,
2-D standard / general form and the 3-D standard form; the default path is untouched (still bitwise the plain oracle);
forms without a contracted variant refuse the flag."""
import zlib

import numpy as np
import pytest

from util import rand2d, rand3d, run_oracle, run_hip_batched

pytestmark = pytest.mark.gpu
C2, FMA, PATH_FUSED = 2, 0x100, 2


def _seed(t):
    return zlib.crc32(repr(t).encode()) % 100000


def _uniform(p, which):
    q = dict(p)
    q['coefs'] = [np.ascontiguousarray(np.broadcast_to(c[..., :1], c.shape)) if k in which else c
                  for k, c in enumerate(p['coefs'])]
    return q


def _same(S, fl, So, flo, what):
    assert np.array_equal(S, So), '%s: %d points differ' % (what, (S != So).sum())
    assert fl[2] == flo[2] and fl[0] == flo[0] and abs(fl[1] - flo[1]) <= 1e-12, (what, fl, flo)


@pytest.mark.parametrize('kind', ['std2d', 'gen2d'])
@pytest.mark.parametrize('BCy,BCx', [('fixed', 'fixed'), ('fixed', 'periodic'), ('extend', 'periodic'), ('extend', 'fixed')])
@pytest.mark.parametrize('shape', [(40, 300), (33, 258), (64, 512), (90, 250), (200, 1200), (17, 24)])
def test_fma_2d_bitwise_against_the_fma_oracle(kind, BCy, BCx, shape):
    yc, xc = shape
    which = (0, 2) if kind == 'std2d' else (0, 2, 3, 4, 5)
    ps = [_uniform(rand2d(kind, yc, xc, BCy, BCx, 0, m & 1, seed=_seed((kind, BCy, BCx, shape, m))), which) for m in range(3)]
    for mx, tol in ((26, 0.0), (200, 2e-4)):
        ref = [run_oracle(p, mx, tol, C2 | FMA) for p in ps]
        plain = [run_oracle(p, mx, tol, C2) for p in ps]
        for kw in (dict(), dict(no_pipe=1), dict(sweeps_per_launch=2), dict(force_tile_skip=1), dict(rows_per_tile=16)):
            S, fl, st = run_hip_batched(ps, mx, tol, fma=1, **kw)
            assert st['path'] == PATH_FUSED
            for m in range(3):
                _same(S[m], fl[m], ref[m][0], ref[m][1], 'fma %s %r member %d %r' % (kind, shape, m, kw))
        assert any((ref[m][0] != plain[m][0]).any() for m in range(3))         # (a different arithmetic, visibly)
        S, fl, st = run_hip_batched(ps, mx, tol)                                   # the default path is the plain one
        for m in range(3):
            _same(S[m], fl[m], plain[m][0], plain[m][1], 'default %s %r member %d' % (kind, shape, m))


@pytest.mark.parametrize('shape', [(9, 20, 66), (12, 33, 130), (50, 40, 250), (21, 50, 240)])
@pytest.mark.parametrize('BCx', ['fixed', 'periodic'])
def test_fma_3d_bitwise_against_the_fma_oracle(shape, BCx):
    zc, yc, xc = shape
    for BCy in ('fixed', 'extend'):
        for msk in (0, 1):
            p = _uniform(rand3d(zc, yc, xc, BCy, BCx, msk, seed=_seed((shape, BCx, BCy, msk))), (0, 1, 2))
            for nsw in (7, 8):
                So, flo = run_oracle(p, nsw - 1, 0.0, C2 | FMA)
                for kw in (dict(), dict(sweeps_per_launch=1), dict(rows_per_tile=8)):
                    S, fl, st = run_hip_batched([p], nsw - 1, 0.0, fma=1, **kw)
                    assert st['path'] == PATH_FUSED and st['xuniform_mask'] == 7
                    _same(S[0], fl[0], So, flo, 'fma 3d %r %s %s %r' % (shape, BCy, BCx, kw))
    ps = [_uniform(rand3d(10, 24, 64, 'fixed', 'periodic', 1, seed=40 + s), (0, 1, 2)) for s in range(4)]
    S, fl, st = run_hip_batched(ps, 300, 2e-3, fma=1)
    for m, q in enumerate(ps):
        So, flo = run_oracle(q, 300, 2e-3, C2 | FMA)
        _same(S[m], fl[m], So, flo, 'fma 3d stop member %d' % m)


def test_fma_is_refused_where_no_contracted_variant_exists():
    from xinvert_amd import _lib
    cases = [rand2d('std2d', 30, 64, 'fixed', 'fixed', 0, 0, seed=1),                      # full coefficient arrays
             rand2d('gen2d', 30, 64, 'fixed', 'fixed', 1, 0, seed=2),                      # B != 0
             _uniform(rand2d('std2d', 30, 65, 'fixed', 'periodic', 0, 0, seed=3), (0, 2)),  # odd-xc periodic seam
             rand3d(6, 12, 20, 'fixed', 'fixed', 0, seed=4)]                               # 3-D, full arrays
    for p in cases:
        with pytest.raises(_lib.XinvError, match='XINV_FLAG_FMA'):
            run_hip_batched([p], 5, 0.0, fma=1)
