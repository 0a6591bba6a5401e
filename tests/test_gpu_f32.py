"""float32 host arrays (xinv_options.f32_mask): uploaded as float32, promoted on the device -- the SAME float64 values a
host-side promotion gives, so the solve is bit for bit the float64-call's -- and, with bit 0, S written back as float32
(the float64 result rounded to nearest).  Chunked uploads, per-member and shared arrays, row-constant coefficients,
strided batches, the front end (a float32 forcing travels as float32 by itself)."""
import ctypes

import numpy as np
import pytest

import util
from util import rand2d, rand3d

pytestmark = pytest.mark.gpu


def _call(ps, mx, tol, f32_mask, shared=(), S_f32=False, **opt):
    """util.run_hip_batched with chosen arrays handed over as float32 (values representable in float32)."""
    from xinvert_amd import _lib
    L = _lib.require_gpu()
    p = ps[0]
    nb = len(ps)
    n = int(np.prod(p['S0'].shape))
    dt = lambda arr: np.float32 if (f32_mask >> arr) & 1 else np.float64
    S = np.ascontiguousarray(np.stack([q['S0'] for q in ps]), dtype=dt(0))
    arrs, strides = [S], [n]
    for k in range(len(p['coefs'])):
        if k in shared:
            arrs.append(np.ascontiguousarray(p['coefs'][k], dtype=dt(k + 1))); strides.append(0)
        else:
            arrs.append(np.ascontiguousarray(np.stack([q['coefs'][k] for q in ps]), dtype=dt(k + 1))); strides.append(n)
    fl = np.tile(np.array([0., 1., 0.]), (nb, 1))
    o = _lib.options(f32_mask=f32_mask, **opt)
    rc = getattr(L, util._FN[p['kind']] + '_batched')(*[_lib.hptr(a, f32=bool((f32_mask >> k) & 1)) for k, a in enumerate(arrs)], nb, _lib.strides_arg(strides),
                                                    *util._scal(p, fl, mx, tol), ctypes.byref(o))
    _lib.check(rc)
    return S, fl, _lib.last_stats()


def _round32(ps):
    """the same problems with every array rounded to float32 precision (kept as float64: the reference call)"""
    out = []
    for p in ps:
        q = dict(p)
        q['S0'] = p['S0'].astype(np.float32).astype(np.float64)
        q['coefs'] = [np.asarray(c).astype(np.float32).astype(np.float64) for c in p['coefs']]
        out.append(q)
    return out


@pytest.mark.parametrize('kind', ['std2d', 'gen2d', 'std3d'])
@pytest.mark.parametrize('chunk', [0, 1, 2])
def test_f32_arrays_equal_the_promoted_float64_call(kind, chunk):
    if kind == 'std3d':
        ps = _round32([rand3d(9, 40, 130, 'fixed', 'periodic', m & 1, seed=60 + m) for m in range(5)])
    else:
        ps = _round32([rand2d(kind, 300, 700, 'extend', 'periodic', 0, m & 1, seed=70 + m) for m in range(5)])
    ncoef = len(ps[0]['coefs'])
    S64, f64, _ = util.run_hip_batched(ps, 40, 1e-7, host_chunk=chunk)
    every = (2 << ncoef) - 1
    forcing_only = 2 << (ncoef - 1)
    for mask in (forcing_only, every & ~1, every):
        S, fl, st = _call(ps, 40, 1e-7, mask, host_chunk=chunk)
        assert np.array_equal(fl, f64), (mask, fl, f64)
        if mask & 1:
            assert S.dtype == np.float32 and np.array_equal(S, S64.astype(np.float32)), mask
        else:
            assert np.array_equal(S, S64), mask


def test_f32_shared_rowconst_and_strided_members():
    """Shared coefficient arrays as float32, one of them one value per row (rowconst_mask), and a batch whose members
    lie 1.5 slices apart in the caller's float32 forcing."""
    from xinvert_amd import _lib
    ps = _round32([rand2d('std2d', 200, 640, 'fixed', 'periodic', 0, 1, seed=90 + m) for m in range(3)])
    for p in ps:
        p['coefs'][0] = np.repeat(ps[0]['coefs'][0][:, :1], 640, axis=1)
        p['coefs'][1] = ps[0]['coefs'][1]; p['coefs'][2] = ps[0]['coefs'][2]
    S64, f64, _ = util.run_hip_batched(ps, 30, 0.0, shared=(0, 1, 2))
    L = _lib.require_gpu()
    n = 200 * 640
    pad = n + n // 2
    Fbuf = np.zeros(3 * pad, dtype=np.float32)
    for m in range(3):
        Fbuf[m * pad:m * pad + n] = ps[m]['coefs'][3].ravel()
    S = np.ascontiguousarray(np.stack([q['S0'] for q in ps]))
    Arow = np.ascontiguousarray(ps[0]['coefs'][0][:, 0], dtype=np.float32)
    B = np.ascontiguousarray(ps[0]['coefs'][1], dtype=np.float32)
    C = np.ascontiguousarray(ps[0]['coefs'][2], dtype=np.float32)
    fl = np.tile(np.array([0., 1., 0.]), (3, 1))
    o = _lib.options(f32_mask=0b11110, rowconst_mask=1)
    rc = L.xinv_standard_2d_f64_batched(_lib.hptr(S), _lib.hptr(Arow, f32=True), _lib.hptr(B, f32=True), _lib.hptr(C, f32=True), _lib.hptr(Fbuf, f32=True), 3,
                                        _lib.strides_arg([n, 0, 0, 0, pad]), *util._scal(ps[0], fl, 30, 0.0), ctypes.byref(o))
    _lib.check(rc)
    assert np.array_equal(S, S64) and np.array_equal(fl, f64)


def test_front_end_sends_a_float32_forcing_as_float32():
    """invert_Poisson on a float32 vorticity: the same field as with the float64 copy of it (the promotion is exact),
    and with iParams['float32_out'] the float32 rounding of it."""
    import xinvert_amd as xa
    rng = np.random.default_rng(5)
    lat = np.linspace(-89.5, 89.5, 180); lon = np.arange(360.0)
    v32 = (1e-5 * rng.standard_normal((3, 180, 360))).astype(np.float32)
    v32[:, 60:80, 100:140] = np.nan
    ip = {'BCs': ['fixed', 'periodic'], 'tolerance': 1e-9, 'mxLoop': 300, 'printInfo': False}
    out = {}
    for tag, arr, extra in (('f64', v32.astype(np.float64), {}), ('f32', v32, {}), ('f32out', v32, {'float32_out': True})):
        F = xa.Field(arr, ('time', 'lat', 'lon'), {'lat': lat, 'lon': lon})
        q = dict(ip); q.update(extra)
        out[tag] = np.asarray(xa.invert_Poisson(F, dims=['lat', 'lon'], coords='lat-lon', iParams=q).values)
    assert out['f32'].dtype == np.float64 and np.array_equal(out['f32'], out['f64'], equal_nan=True)
    assert out['f32out'].dtype == np.float32 and np.array_equal(out['f32out'], out['f64'].astype(np.float32), equal_nan=True)
