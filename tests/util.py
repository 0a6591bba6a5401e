"""Shared helpers for the parity tests: problem builders, oracle and HIP runners."""
import ctypes
import os

import numpy as np

U = -9.99e8
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


# ------------------------------------------------------------------ problem builders
def rand2d(kind, yc, xc, BCy, BCx, bnz=False, msk=False, seed=0, dely=1.3, delx=1.1, omega=1.3):
    """Random well-conditioned 2-D problem (same recipe as tests/golden/gen_golden.py)."""
    rng = np.random.default_rng(seed)
    mk = lambda: rng.uniform(0.5, 1.5, (yc, xc))
    A, C = mk(), mk()
    B = rng.uniform(-.2, .2, (yc, xc)) if bnz else np.zeros((yc, xc))
    F = rng.standard_normal((yc, xc))
    if msk:
        F[rng.random((yc, xc)) < 0.15] = U
        A[rng.random((yc, xc)) < 0.03] = U
        if bnz:
            B[rng.random((yc, xc)) < 0.03] = U
    S0 = rng.standard_normal((yc, xc)) * 0.1
    if msk:
        S0[rng.random((yc, xc)) < 0.05] = U
    r = delx / dely
    p = dict(kind=kind, yc=yc, xc=xc, BCy=BCy, BCx=BCx, dely=dely, delx=delx, delxSqr=delx**2,
             ratio=r, ratioQtr=r / 4, ratioSqr=r**2, optArg=omega, undef=U, S0=S0)
    if kind == 'std2d':
        p['coefs'] = [A, B, C, F]
    else:
        D, E = mk() * 0.1, mk() * 0.1
        Fc = -mk() * 0.01
        p['coefs'] = [A, B, C, D, E, Fc, F]
    return p


def rand2dt(yc, xc, BCy, BCx, bnz=False, msk=False, seed=0, dely=1.3, delx=1.1, omega=1.3):
    """Random problem for the standard 2-D "test" form (A, B, C, D, E, F)."""
    rng = np.random.default_rng(seed)
    sh = (yc, xc)
    mk = lambda s=1.0: rng.uniform(0.5, 1.5, sh) * s
    A, D = mk(), mk()
    B = rng.uniform(-.2, .2, sh) if bnz else np.zeros(sh)
    C = rng.uniform(-.2, .2, sh) if bnz else np.zeros(sh)
    E = -mk(0.05)
    F = rng.standard_normal(sh)
    if msk:
        F[rng.random(sh) < 0.15] = U
        A[rng.random(sh) < 0.03] = U
    S0 = rng.standard_normal(sh) * 0.1
    if msk:
        S0[rng.random(sh) < 0.05] = U
    r = delx / dely
    return dict(kind='std2dt', yc=yc, xc=xc, BCy=BCy, BCx=BCx, dely=dely, delx=delx, delxSqr=delx**2,
                ratio=r, ratioQtr=r / 4, ratioSqr=r**2, optArg=omega, undef=U, S0=S0,
                coefs=[A, B, C, D, E, F])


def randbih(yc, xc, BCy, BCx, bnz=False, msk=False, seed=0, dely=1.3, delx=1.1, omega=0.9):
    """Random biharmonic problem (10 coefficient arrays A..J)."""
    rng = np.random.default_rng(seed)
    sh = (yc, xc)
    mk = lambda s=1.0: rng.uniform(0.5, 1.5, sh) * s
    A, C = mk(), mk()
    B = mk(0.3) if bnz else np.zeros(sh)
    D, E, F = -mk(0.5), (mk(0.1) if bnz else np.zeros(sh)), -mk(0.5)
    G, H, I = mk(0.05), mk(0.05), mk(0.01)
    J = rng.standard_normal(sh)
    if msk:
        J[rng.random(sh) < 0.15] = U
        A[rng.random(sh) < 0.03] = U
    S0 = rng.standard_normal(sh) * 0.1
    if msk:
        S0[rng.random(sh) < 0.05] = U
    r = delx / dely
    return dict(kind='bih2d', yc=yc, xc=xc, BCy=BCy, BCx=BCx, dely=dely, delx=delx,
                delxSSr=delx**4, delxTr=delx**3, delxSqr=delx**2, ratio=r, ratioSSr=r**4,
                ratioQtr=r / 4, ratioSqr=r**2, optArg=omega, undef=U, S0=S0,
                coefs=[A, B, C, D, E, F, G, H, I, J])


def rand3d(zc, yc, xc, BCy, BCx, msk=False, seed=0, delz=2.0, dely=1.3, delx=1.1, omega=1.2):
    rng = np.random.default_rng(seed)
    sh = (zc, yc, xc)
    mk = lambda: rng.uniform(0.5, 1.5, sh)
    A, B, C = mk(), mk(), mk()
    F = rng.standard_normal(sh)
    if msk:
        F[rng.random(sh) < 0.15] = U
        B[rng.random(sh) < 0.03] = U
    S0 = rng.standard_normal(sh) * 0.1
    if msk:
        S0[rng.random(sh) < 0.05] = U
    return dict(kind='std3d', zc=zc, yc=yc, xc=xc, BCz='fixed', BCy=BCy, BCx=BCx, delz=delz,
                dely=dely, delx=delx, delxSqr=delx**2, ratio2Sqr=(delx / delz)**2,
                ratio1Sqr=(delx / dely)**2, optArg=omega, undef=U, S0=S0, coefs=[A, B, C, F])


def rand3dg(zc, yc, xc, BCy, BCx, msk=False, seed=0, delz=2.0, dely=1.3, delx=1.1, omega=1.2):
    """Random problem for the general 3-D form (A..G, forcing H)."""
    rng = np.random.default_rng(seed)
    sh = (zc, yc, xc)
    mk = lambda s=1.0: rng.uniform(0.5, 1.5, sh) * s
    A, B, C = mk(), mk(), mk()
    D, E, F = mk(0.1), -mk(0.1), mk(0.1)
    G = -mk(0.01)
    H = rng.standard_normal(sh)
    if msk:
        H[rng.random(sh) < 0.15] = U
        H[:, :, 0][rng.random((zc, yc)) < 0.3] = U
        B[rng.random(sh) < 0.03] = U
        G[rng.random(sh) < 0.03] = U
    S0 = rng.standard_normal(sh) * 0.1
    if msk:
        S0[rng.random(sh) < 0.05] = U
    return dict(kind='gen3d', zc=zc, yc=yc, xc=xc, BCz='fixed', BCy=BCy, BCx=BCx, delz=delz,
                dely=dely, delx=delx, delxSqr=delx**2, ratio2=delx / delz, ratio1=delx / dely,
                ratio2Sqr=(delx / delz)**2, ratio1Sqr=(delx / dely)**2, optArg=omega, undef=U,
                S0=S0, coefs=[A, B, C, D, E, F, G, H])


def eliassen_problem(tag):
    """The reference's Eliassen tests (tests/test_Eliassen.py:16-147 Hadley, :207-232 TC) through
    the front end's coefficient code: real data, true 9-point form (B != 0)."""
    from xinvert_amd import apps
    from xinvert_amd.field import Field
    d = golden('eliassen.npz')
    lev, lat = d[tag + '_lev'], d[tag + '_lat']
    F = Field(d[tag + '_F'], ('lev', 'lat'), {'lev': lev, 'lat': lat})
    iP = apps._update(apps.default_iParams, {'BCs': ['fixed', 'fixed']})
    mP = apps._update(apps.default_mParams, {k: d[tag + '_' + k] for k in 'ABC'},
                      ['A', 'B', 'C', 'g', 'Omega', 'Rearth'])
    Fm, initS, cs = apps._coeffs_Eliassen(F, ['lev', 'lat'], 'z-lat', mP, iP, None)
    ps = apps._cal_params2D(lev, lat, 'z-lat')
    yc, xc = F.shape
    p = dict(kind='std2d', yc=yc, xc=xc, BCy='fixed', BCx='fixed', dely=ps['del2'], delx=ps['del1'],
             delxSqr=ps['del1Sqr'], ratioQtr=ps['ratioQtr'], ratioSqr=ps['ratioSqr'],
             optArg=float(d[tag + '_optArg']), undef=U, S0=np.zeros((yc, xc)),
             coefs=[np.ascontiguousarray(c) for c in cs] + [np.ascontiguousarray(Fm.values)])
    return p, d, ps


# ------------------------------------------------------------------ oracle runner
def run_oracle(p, mxLoop, tol, order):
    import oracle as orc
    S = np.array(p['S0'], dtype=np.float64, copy=True)
    fl = np.array([0., 1., 0.])
    c = [np.ascontiguousarray(a, dtype=np.float64) for a in p['coefs']]
    if p['kind'] == 'std2d':
        orc.standard_2d(S, *c, p['yc'], p['xc'], p['dely'], p['delx'], p['BCy'], p['BCx'],
                        p['delxSqr'], p['ratioQtr'], p['ratioSqr'], p['optArg'], p['undef'], fl,
                        mxLoop, tol, order)
    elif p['kind'] == 'gen2d':
        orc.general_2d(S, *c, p['yc'], p['xc'], p['dely'], p['delx'], p['BCy'], p['BCx'],
                       p['delxSqr'], p['ratio'], p['ratioQtr'], p['ratioSqr'], p['optArg'],
                       p['undef'], fl, mxLoop, tol, order)
    elif p['kind'] == 'std2dt':
        orc.standard_2d_test(S, *c, p['yc'], p['xc'], p['dely'], p['delx'], p['BCy'], p['BCx'],
                             p['delxSqr'], p['ratioQtr'], p['ratioSqr'], p['optArg'], p['undef'], fl,
                             mxLoop, tol, order)
    elif p['kind'] == 'bih2d':
        orc.general_bih_2d(S, *c, p['yc'], p['xc'], p['dely'], p['delx'], p['BCy'], p['BCx'],
                           p['delxSSr'], p['delxTr'], p['delxSqr'], p['ratio'], p['ratioSSr'],
                           p['ratioQtr'], p['ratioSqr'], p['optArg'], p['undef'], fl, mxLoop, tol,
                           order)
    elif p['kind'] == 'gen3d':
        orc.general_3d(S, *c, p['zc'], p['yc'], p['xc'], p['delz'], p['dely'], p['delx'],
                       p['BCz'], p['BCy'], p['BCx'], p['delxSqr'], p['ratio2'], p['ratio1'],
                       p['ratio2Sqr'], p['ratio1Sqr'], p['optArg'], p['undef'], fl, mxLoop, tol, order)
    else:
        orc.standard_3d(S, *c, p['zc'], p['yc'], p['xc'], p['delz'], p['dely'], p['delx'],
                        p['BCz'], p['BCy'], p['BCx'], p['delxSqr'], p['ratio2Sqr'], p['ratio1Sqr'],
                        p['optArg'], p['undef'], fl, mxLoop, tol, order)
    return S, fl


# ------------------------------------------------------------------ HIP runners (C-ABI)
def _scal(p, flags, mxLoop, tol):
    from xinvert_amd import _lib
    b = _lib.bc
    fp = _lib.hptr(flags)
    if p['kind'] in ('std2d', 'std2dt'):
        return [p['yc'], p['xc'], p['dely'], p['delx'], b(p['BCy']), b(p['BCx']), p['delxSqr'],
                p['ratioQtr'], p['ratioSqr'], p['optArg'], p['undef'], fp, mxLoop, tol]
    if p['kind'] == 'gen2d':
        return [p['yc'], p['xc'], p['dely'], p['delx'], b(p['BCy']), b(p['BCx']), p['delxSqr'],
                p['ratio'], p['ratioQtr'], p['ratioSqr'], p['optArg'], p['undef'], fp, mxLoop, tol]
    if p['kind'] == 'bih2d':
        return [p['yc'], p['xc'], p['dely'], p['delx'], b(p['BCy']), b(p['BCx']), p['delxSSr'],
                p['delxTr'], p['delxSqr'], p['ratio'], p['ratioSSr'], p['ratioQtr'], p['ratioSqr'],
                p['optArg'], p['undef'], fp, mxLoop, tol]
    if p['kind'] == 'gen3d':
        return [p['zc'], p['yc'], p['xc'], p['delz'], p['dely'], p['delx'], b(p['BCz']), b(p['BCy']),
                b(p['BCx']), p['delxSqr'], p['ratio2'], p['ratio1'], p['ratio2Sqr'], p['ratio1Sqr'],
                p['optArg'], p['undef'], fp, mxLoop, tol]
    return [p['zc'], p['yc'], p['xc'], p['delz'], p['dely'], p['delx'], b(p['BCz']), b(p['BCy']),
            b(p['BCx']), p['delxSqr'], p['ratio2Sqr'], p['ratio1Sqr'], p['optArg'], p['undef'],
            fp, mxLoop, tol]


_FN = {'std2d': 'xinv_standard_2d_f64', 'gen2d': 'xinv_general_2d_f64', 'std3d': 'xinv_standard_3d_f64',
       'bih2d': 'xinv_general_bih_2d_f64', 'std2dt': 'xinv_standard_2d_test_f64',
       'gen3d': 'xinv_general_3d_f64'}


def run_hip_single(p, mxLoop, tol):
    """The positional single-slice twin of the numba kernel (host pointers)."""
    from xinvert_amd import _lib
    L = _lib.require_gpu()
    S = np.array(p['S0'], dtype=np.float64, copy=True)
    fl = np.array([0., 1., 0.])
    c = [np.ascontiguousarray(a, dtype=np.float64) for a in p['coefs']]
    rc = getattr(L, _FN[p['kind']])(_lib.hptr(S), *[_lib.hptr(a) for a in c], *_scal(p, fl, mxLoop, tol))
    _lib.check(rc)
    return S, fl


def run_hip_batched(ps, mxLoop, tol, shared=(), **opt):
    """Host-pointer batched entry.  `ps`: list of problems of identical geometry; coefficient
    indices in `shared` are passed once with batch stride 0."""
    from xinvert_amd import _lib
    L = _lib.require_gpu()
    p = ps[0]
    nb = len(ps)
    n = int(np.prod(p['S0'].shape))
    S = np.ascontiguousarray(np.stack([q['S0'] for q in ps]), dtype=np.float64)
    arrs, strides = [S], [n]
    for k in range(len(p['coefs'])):
        if k in shared:
            arrs.append(np.ascontiguousarray(p['coefs'][k], dtype=np.float64)); strides.append(0)
        else:
            arrs.append(np.ascontiguousarray(np.stack([q['coefs'][k] for q in ps]), dtype=np.float64))
            strides.append(n)
    fl = np.tile(np.array([0., 1., 0.]), (nb, 1))
    o = _lib.options(**opt)
    rc = getattr(L, _FN[p['kind']] + '_batched')(*[_lib.hptr(a) for a in arrs], nb,
                                                 _lib.strides_arg(strides),
                                                 *_scal(p, fl, mxLoop, tol), ctypes.byref(o))
    _lib.check(rc)
    return S, fl, _lib.last_stats()


def run_hip_dev(ps, mxLoop, tol, shared=(), stream=None, **opt):
    """Device-pointer batched entry with torch-owned HBM buffers."""
    import torch
    from xinvert_amd import _lib
    L = _lib.require_gpu()
    p = ps[0]
    nb = len(ps)
    n = int(np.prod(p['S0'].shape))
    dev = torch.device('cuda', 0)
    S = torch.from_numpy(np.ascontiguousarray(np.stack([q['S0'] for q in ps]), dtype=np.float64)).to(dev)
    ts, strides = [S], [n]
    for k in range(len(p['coefs'])):
        if k in shared:
            ts.append(torch.from_numpy(np.ascontiguousarray(p['coefs'][k], dtype=np.float64)).to(dev))
            strides.append(0)
        else:
            ts.append(torch.from_numpy(np.ascontiguousarray(
                np.stack([q['coefs'][k] for q in ps]), dtype=np.float64)).to(dev))
            strides.append(n)
    torch.cuda.synchronize()
    fl = np.tile(np.array([0., 1., 0.]), (nb, 1))
    o = _lib.options(**opt)
    sp = ctypes.c_void_p(stream.cuda_stream) if stream is not None else None
    rc = getattr(L, _FN[p['kind']] + '_dev')(*[ctypes.c_void_p(t.data_ptr()) for t in ts], nb,
                                             _lib.strides_arg(strides),
                                             *_scal(p, fl, mxLoop, tol), ctypes.byref(o), sp)
    _lib.check(rc)
    return S.cpu().numpy(), fl, _lib.last_stats()


def rel_l2(a, b, mask=None):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    if mask is not None:
        a, b = a[mask], b[mask]
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


# ------------------------------------------------------------------ a minimal `xarray` for the front end's edge
class _StandinCoord:
    def __init__(self, v):
        self.values = np.asarray(v)


class StandinDataArray:
    """What xinvert_amd.field reads of an xarray.DataArray (from_any) and how it builds one (to_like): `values`, `dims`,
    `coords[d].values`, `name`, and the constructor `DataArray(values, dims=..., coords=..., name=...)`."""

    def __init__(self, values, dims=None, coords=None, name=None):
        self.values = np.asarray(values)
        self.dims = tuple(dims)
        self.coords = {d: (c if hasattr(c, 'values') else _StandinCoord(c)) for d, c in (coords or {}).items()}
        self.name = name


class xarray_standin:
    """Context manager: `import xarray` finds a module whose DataArray is StandinDataArray (xarray is not installed in the
    build image; the real-xarray return branch of field.to_like would otherwise never execute anywhere)."""

    def __enter__(self):
        import sys
        import types
        self.prev = sys.modules.get('xarray')
        m = types.ModuleType('xarray')
        m.DataArray = StandinDataArray
        sys.modules['xarray'] = m
        return m

    def __exit__(self, *exc):
        import sys
        if self.prev is None:
            sys.modules.pop('xarray', None)
        else:
            sys.modules['xarray'] = self.prev
        return False


def p3_extend_ok(yc):
    """The two-sweep 3-D pass takes BCy = 'extend' at every row count since the row blocks may be shifted up by two rows
    (xinv_tiles.h: xinv_p3_extend_joff); kept for the tests that asked which counts it took before."""
    return True
