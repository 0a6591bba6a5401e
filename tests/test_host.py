"""Host logic (no GPU): the front-end restatement, the boundary wrappers' marshalling and
error behaviour, and the C-ABI library's exports.  No compute call is made here."""
import os
import re

import numpy as np
import pytest

from xinvert_amd import _lib, apps, core
from xinvert_amd.field import Field

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------ C-ABI
def test_library_loads_and_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, 'include', 'xinv.h')).read()
    declared = set(re.findall(r'\b(xinv_[a-z0-9_]+)\s*\(', hdr))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    L = _lib.load()
    for name in declared:
        assert getattr(L, name) is not None
    assert L.xinv_version() >= 100
    # the hand-written ctypes mirrors of xinv_options / xinv_stats have the library's sizes (load() enforces it too)
    import ctypes
    so, ss = ctypes.c_int32(0), ctypes.c_int32(0)
    L.xinv_abi_sizes(ctypes.byref(so), ctypes.byref(ss))
    assert (so.value, ss.value) == (ctypes.sizeof(_lib.XinvOptions), ctypes.sizeof(_lib.XinvStats))


def test_shipped_library_does_not_import_getenv():
    """no environment switch alters a production solve -- the shipped shared object does not even import
    getenv (the planner's overrides are xinv_options fields; only the test-hooks / experiment builds read XINV_*)."""
    import shutil
    import subprocess
    from xinvert_amd import _lib
    nm = shutil.which('nm') or '/opt/rocm/lib/llvm/bin/llvm-nm'
    out = subprocess.run([nm, '-D', '--undefined-only', _lib.SO], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert 'hipLaunchKernel' in out.stdout or 'hipModuleLaunchKernel' in out.stdout     # (the listing is the real import table)
    assert 'getenv' not in out.stdout


def test_no_cpu_fallback_without_gpu():
    L = _lib.load()
    if L.xinv_device_count() > 0:
        pytest.skip('a GPU is visible')
    with pytest.raises(_lib.XinvError):
        _lib.require_gpu()
    F = Field(np.zeros((5, 8)), ('lat', 'lon'), {'lat': np.linspace(-10, 10, 5), 'lon': np.linspace(0, 70, 8)})
    with pytest.raises(_lib.XinvError):
        apps.invert_Poisson(F, ['lat', 'lon'], iParams={'printInfo': False})
    # the host-pointer entry points report the missing device instead of computing anything
    S = np.zeros((5, 8)); fl = np.array([0., 1., 0.])
    rc = L.xinv_standard_2d_f64(_lib.hptr(S), _lib.hptr(S), None, _lib.hptr(S), _lib.hptr(S),
                                5, 8, 1., 1., 0, 0, 1., .25, 1., 1., -9.99e8, _lib.hptr(fl), 3, 1e-8)
    assert rc == -3 and b'no HIP device' in L.xinv_last_error()


def test_product_never_imports_the_oracle():
    for root, _, files in os.walk(os.path.join(ROOT, 'xinvert_amd')):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                txt = open(os.path.join(root, f)).read()
                assert 'import oracle' not in txt and 'from oracle' not in txt, f
                assert 'xinv_oracle' not in txt, f


# ------------------------------------------------------------------ parameters (a8)
OMEGA_2D = {(73, 144): 1.934805014448645, (180, 360): 1.972913967488753,
            (1800, 3600): 1.997245570836297, (2000, 2000): 1.996864900786959,
            (720, 1440): 1.993133264690061, (151, 201): 1.964078724984458}


def test_optimal_omega_goldens():
    for (gy, gx), w in OMEGA_2D.items():
        p = apps._cal_params2D(np.linspace(0, 1, gy), np.linspace(0, 1, gx), 'cartesian')
        assert abs(p['optArg'] - w) < 2e-15
        assert 1.0 <= p['optArg'] <= 2.0                 # reference tests/test_OptArg.py:24
    p = apps._cal_params3D(np.linspace(0, 1, 50), np.linspace(0, 1, 360), np.linspace(0, 1, 720), 'cartesian')
    assert abs(p['optArg'] - 1.916326873236605) < 2e-15  # uses 2*gc3+3 (apps.py:2208)


def test_cal_params2d_metrics():
    lat = np.linspace(-90, 90, 73); lon = np.linspace(0, 357.5, 144)
    p = apps._cal_params2D(lat, lon, 'lat-lon')
    R = 6371200.0
    assert p['del2'] == np.deg2rad(2.5) * R and p['del1'] == np.deg2rad(2.5) * R
    assert p['ratio'] == 1.0 and p['ratioQtr'] == 0.25 and p['del1Sqr'] == p['del1'] ** 2.0
    assert list(p['flags']) == [0.0, 1.0, 0.0]
    q = apps._cal_params2D(np.linspace(0, 10, 11), lat, 'z-lat')
    assert q['del2'] == 1.0 and q['del1'] == np.deg2rad(2.5) * R
    with pytest.raises(Exception, match='unsupported coords'):
        apps._cal_params2D(lat, lon, 'polar')
    with pytest.raises(Exception, match='non-uniform'):
        apps._cal_params2D(np.array([0., 1., 3.]), lon, 'cartesian')


def test_update_semantics():
    d = apps._update(apps.default_iParams, {'optArg': None, 'mxLoop': 7})
    assert d['optArg'] is None and d['mxLoop'] == 7 and d['tolerance'] == 1e-8
    with pytest.raises(Exception, match='is not used'):
        apps._update(apps.default_mParams, {'bogus': 1}, ['g'])
    ps = {'optArg': 1.7, 'mxLoop': 1}
    assert apps._update(ps, d)['optArg'] == 1.7          # user None does not override computed
    assert apps._update(ps, d)['mxLoop'] == 7


def test_mask_fs_nan_and_value_undef():
    v = np.array([[1.0, np.nan, 3.0], [4.0, 5.0, -9999.0], [7.0, 8.0, 9.0]])
    F = Field(v, ('y', 'x'))
    m, s, z = apps._mask_FS(F, ['y', 'x'], {'undef': np.nan, 'BCs': ['fixed', 'fixed']}, None)
    assert m.values[0, 1] == -9.99e8 and m.values[1, 2] == -9999.0 and (s.values == 0).all()
    m, s, z = apps._mask_FS(F, ['y', 'x'], {'undef': -9999.0, 'BCs': ['fixed', 'fixed']}, None)
    assert m.values[1, 2] == -9.99e8 and np.isnan(m.values[0, 1])
    # icbc: kept on masked points and on the first/last index of non-periodic dims (apps.py:2144-2156)
    ic = Field(np.full((3, 3), 2.5), ('y', 'x'))
    m, s, z = apps._mask_FS(F, ['y', 'x'], {'undef': np.nan, 'BCs': ['fixed', 'periodic']}, ic)
    assert (s.values[0] == 2.5).all() and (s.values[2] == 2.5).all()
    assert s.values[1, 0] == 0 and s.values[1, 1] == 0


def test_poisson_coefficients_latlon():
    lat = np.linspace(-80, 80, 9); lon = np.linspace(0, 315, 8)
    z = np.ones((2, 9, 8)); z[1, 4, 3] = np.nan
    F = Field(z, ('time', 'lat', 'lon'), {'lat': lat, 'lon': lon})
    Fm, S, (A, B, C) = apps._coeffs_Poisson(F, ['lat', 'lon'], 'lat-lon', apps.default_mParams,
                                            apps.default_iParams, None)
    assert A.shape == (9, 8) and np.isnan(A[0]).all()            # shifted half grid: row 0 never read
    assert np.allclose(A[1], np.cos(np.deg2rad((lat[1] + lat[0]) / 2)))
    assert (B == 0).all() and np.allclose(C[:, 0], 1 / np.cos(np.deg2rad(lat)))
    assert Fm.values[1, 4, 3] == -9.99e8 and Fm.values[0, 4, 3] == np.cos(np.deg2rad(lat[4]))


def test_remaining_app_coefficients_follow_the_reference_formulas():
    """Spot values of the coefficient builders added for SURVEY 8(f) rank 4, written out from
    the reference's formulas (apps.py:1440-1467, 1556-1606, 1660-1709, 1751-1790, 1839-1931,
    2055-2109) independently of the builders."""
    Om, Re = 7.292e-5, 6371200.0
    lat = np.linspace(-30, 40, 8); lon = np.linspace(0, 315, 8)
    z = np.ones((2, 8, 8)); z[1, 4, 3] = np.nan
    F = Field(z, ('time', 'lat', 'lon'), {'lat': lat, 'lon': lon})
    iP = apps._update(apps.default_iParams, {})
    mP = apps._update(apps.default_mParams, {'epsilon': 1e-5, 'Phi': 5000., 'D': 200., 'R': 1e-4})
    la = np.deg2rad(lat); f = 2 * Om * np.sin(la)
    j = 5
    half = (la[j] + la[j - 1]) / 2
    # Gill-Matsuno, flux form
    Fm, S, (A, B, C, D, E) = apps._coeffs_GillMatsuno_test(F, ['lat', 'lon'], 'lat-lon', mP, iP, None)
    fH = 2 * Om * np.sin(half)
    assert np.isnan(A[0, 0, 0]) and A[1, j, 2] == 1e-5 / (1e-10 + fH**2) * 5000. * np.cos(half)
    assert B[0, j, 0] == -(f[j] / (1e-10 + f[j]**2)) * 5000. and C[0, j, 0] == -B[0, j, 0]
    assert D[0, j, 1] == 1e-5 / (1e-10 + f[j]**2) * 5000. / np.cos(la[j]) and E[0, j, 1] == -1e-5 * np.cos(la[j])
    assert Fm.values[1, 4, 3] == -9.99e8 and Fm.values[0, j, 3] == np.cos(la[j])
    # Stommel, flux form
    Fm, S, (A, B, C, D, E) = apps._coeffs_Stommel_test(F, ['lat', 'lon'], 'lat-lon', mP, iP, None)
    assert A[0, j, 0] == -1e-4 / 200. * np.cos(half) and B[0, j, 0] == -f[j] and C[0, j, 0] == f[j]
    assert D[0, j, 0] == -1e-4 / 200. / np.cos(la[j]) and (E == 0).all()
    assert Fm.values[0, j, 0] == -1.0 / 200. / 1027 * np.cos(la[j])
    # Stommel-Arons: the Gill-Matsuno operator with Phi = 1, F == 0
    G, S, (A, B, C, D, E, Fc) = apps._coeffs_StommelArons(F, ['lat', 'lon'], 'lat-lon', mP, iP, None)
    c1 = 1e-5 / (1e-10 + f**2); c2 = f / (1e-10 + f**2); d2m = Re / 180. * np.pi
    assert A[0, j, 0] == c1[j] and C[0, j, 0] == c1[j] / np.cos(la[j])**2 and (B == 0).all() and (Fc == 0).all()
    assert D[0, j, 0] == np.gradient(c1, lat)[j] / d2m + c1[j] * np.tan(la[j]) / Re
    assert E[0, j, 0] == -np.gradient(c2, lat)[j] / d2m / np.cos(la[j])
    # geostrophic: |f| < 2e-5 inflated by 1.5; forcing rebuilt from the RAW input (NaN stays NaN)
    Fm, S, (A, B, C) = apps._coeffs_geostrophic(F, ['lat', 'lon'], 'lat-lon', mP, iP, None)
    jj = int(np.argmin(np.abs(lat)))
    assert abs(f[jj]) < 2e-5 and C[0, jj, 0] == f[jj] * 1.5 / np.cos(la[jj])
    assert C[0, j, 0] == f[j] / np.cos(la[j]) and A[0, j, 0] == fH * np.cos(half)
    assert np.isnan(Fm.values[1, 4, 3])
    # PV2D / RefState / Eliassen in a vertical plane
    lev = np.linspace(1e5, 2e4, 5)
    P = Field(np.full((5, 8), 2e-6), ('lev', 'lat'), {'lev': lev, 'lat': lat + 45.})
    n2 = Field(np.linspace(1e-4, 3e-4, 5), ('lev',), {'lev': lev})
    Fm, S, (A, B, C) = apps._coeffs_PV2D(P, ['lev', 'lat'], 'z-lat', dict(mP, N2=n2), iP, None)
    assert A.shape == (5, 8) and A[3, 2] == mP['f0']**2 / n2.values[3] and (C == 1).all() and (B == 0).all()
    gam = Field(np.linspace(1e-3, 2e-3, 5), ('lev',), {'lev': lev})
    Fm, S, (A, B, C) = apps._coeffs_RefState(P, ['lev', 'lat'], 'z-lat', dict(mP, Gamma=gam), iP, None)
    assert A[2, 3] == np.sin(np.deg2rad(lat[3] + 45.)) and C[2, 3] == gam.values[2] * mP['g'] / 2e-6 / (lat[3] + 45.)
    Fm, S, (A, B, C) = apps._coeffs_RefState(P, ['lev', 'lat'], 'cartesian', dict(mP, Gamma=gam), iP, None)
    assert A[2, 3] == 2.0 * mP['ang0'] / (lat[3] + 45.)**3.0
    with pytest.raises(Exception, match='is not used'):
        apps.invert_RefState(P, ['lev', 'lat'], mParams={'ang0': 1.0, 'Gamma': gam})     # the valid key is 'Ang0'
    a = Field(np.arange(40.).reshape(8, 5), ('lat', 'lev'), {'lev': lev, 'lat': lat + 45.})
    Fm, S, (A, B, C) = apps._coeffs_Eliassen(P, ['lev', 'lat'], 'z-lat', dict(mP, A=a, B=0.5, C=np.ones((5, 8))), iP, None)
    assert A.shape == (5, 8) and A[1, 2] == a.values[2, 1] and (B == 0.5).all() and (C == 1).all()
    # 3-D ocean
    F3 = Field(np.ones((5, 8, 8)), ('lev', 'lat', 'lon'), {'lev': np.linspace(0, 400, 5), 'lat': lat, 'lon': lon})
    n2 = Field(np.linspace(2e-4, 1e-4, 5), ('lev',), {'lev': F3['lev']})
    H, S, (A, B, C, D, E, Fc, G) = apps._coeffs_3DOcean(F3, ['lev', 'lat', 'lon'], 'lat-lon',
                                                         dict(mP, N2=n2, k=1e-7), iP, None)
    c3 = 1e-7 / n2.values
    assert A[3, j, 0] == c3[3] and B[3, j, 0] == c1[j] and C[3, j, 0] == c1[j] / np.cos(la[j])**2
    assert D[3, j, 0] == np.gradient(c3, F3['lev'])[3] and (G == 0).all()
    assert E[0, j, 0] == np.gradient(c1, lat)[j] / d2m - c1[j] * np.tan(la[j]) / Re
    assert Fc[0, j, 0] == -np.gradient(c2, lat)[j] / d2m / np.cos(la[j])


def test_dims_length_errors():
    F = Field(np.zeros((4, 5)), ('y', 'x'))
    with pytest.raises(Exception, match='2 dimensions are needed for inversion'):
        core.inv_standard2D(0, 0, 0, F, F, ['y'], {})
    with pytest.raises(Exception, match='2 dimensions are needed for inversion'):
        core.inv_general2D(0, 0, 0, 0, 0, 0, F, F, ['y', 'x', 'z'], {})
    with pytest.raises(Exception, match='3 dimensions are needed for inversion'):
        core.inv_standard3D(0, 0, 0, F, F, ['y', 'x'], {})
    with pytest.raises(Exception, match='dimensional forcing are needed'):
        apps.invert_Poisson(F, ['y'])
    with pytest.raises(Exception, match='is not used'):
        apps.invert_Poisson(F, ['y', 'x'], mParams={'Phi': 1.0})


def test_omega_validates_stratification():
    F = Field(np.zeros((3, 4, 5)), ('lev', 'lat', 'lon'))
    with pytest.raises(Exception, match='unstable stratification'):
        apps.invert_omega(F, ['lev', 'lat', 'lon'], mParams={'N2': np.array([1e-4, -1e-4, 1e-4])})
    with pytest.raises(Exception, match='inifinite stratification'):
        apps.invert_omega(F, ['lev', 'lat', 'lon'], mParams={'N2': np.array([1e-4, np.inf, 1e-4])})


# ------------------------------------------------------------------ batching / marshalling
def test_batch_layout_and_shared_coefficients():
    F = Field(np.zeros((3, 4, 2, 5)), ('time', 'lat', 'mem', 'lon'))
    perm, bdims, bshape = core._batch_layout(F, ['lat', 'lon'])
    assert perm == [0, 2, 1, 3] and bdims == ['time', 'mem'] and bshape == [3, 2]
    core2 = np.arange(20.0).reshape(4, 5)
    a, st, rc = core._prep_coef(core2, F, perm, (4, 5), 6)
    assert st == 0 and a.shape == (4, 5) and not rc
    full = np.broadcast_to(core2[None, :, None, :], F.shape)      # zero batch strides -> shared
    a, st, rc = core._prep_coef(full, F, perm, (4, 5), 6)
    assert st == 0 and np.array_equal(a, core2) and not rc
    per = np.arange(120.0).reshape(3, 4, 2, 5)
    a, st, rc = core._prep_coef(per, F, perm, (4, 5), 6)
    assert st == 20 and a.shape == (6, 4, 5) and np.array_equal(a[1], per[0, :, 1, :]) and not rc
    with pytest.raises(Exception, match='matches neither'):
        core._prep_coef(np.zeros((7, 7)), F, perm, (4, 5), 6)
    # one value per row (stride-0 view along x): only the rows travel (xinv_options.rowconst_mask)
    lat = np.arange(4.0) + 1
    a, st, rc = core._prep_coef(np.broadcast_to(lat[:, None], (4, 5)), F, perm, (4, 5), 6)
    assert rc and st == 0 and a.shape == (4,) and np.array_equal(a, lat)
    rows_b = np.broadcast_to(np.arange(24.0).reshape(3, 4, 2, 1), F.shape)              # per member, row-constant
    a, st, rc = core._prep_coef(rows_b, F, perm, (4, 5), 6)
    assert rc and st == 4 and a.shape == (6, 4) and np.array_equal(a[1], rows_b[0, :, 1, 0])
    # identically-zero stride-0 view: NULL for the cross coefficient only
    z = np.broadcast_to(np.float64(0.0), (4, 5))
    assert core._prep_coef(z, F, perm, (4, 5), 6, allow_null=True)[0] is None
    a, st, rc = core._prep_coef(z, F, perm, (4, 5), 6)
    assert rc and a.shape == (4,) and not a.any()
    # the lat-lon builders produce such views
    lt = np.linspace(-60, 60, 7); ln = np.linspace(0, 300, 6)
    Fp = Field(np.ones((7, 6)), ('lat', 'lon'), {'lat': lt, 'lon': ln})
    Fm, S0, (A, B, C) = apps._coeffs_Poisson(Fp, ['lat', 'lon'], 'lat-lon', apps.default_mParams,
                                             apps.default_iParams, None)
    assert A.shape == B.shape == C.shape == (7, 6) and A.strides[-1] == 0 and C.strides[-1] == 0
    assert all(st == 0 for st in B.strides)


def test_empty_batch_axis_is_a_no_op():
    """An empty non-core axis: the reference's `for selDict in loop_noncore(...)` body never runs and S
    comes back untouched -- no library call (and so no GPU needed) here either."""
    F = Field(np.zeros((0, 6, 8)), ('time', 'lat', 'lon'), {'lat': np.linspace(-50, 50, 6), 'lon': np.arange(8) * 45.})
    S = apps.invert_Poisson(F, ['lat', 'lon'], iParams={'BCs': ['fixed', 'periodic'], 'printInfo': False})
    assert S.shape == (0, 6, 8)


def test_info_line_format():
    assert core._info({'time': np.float64(3.0)}) == '{time: 3.0}'
    assert core._info({}) == '{}'
    line = core._info({'lev': 500}) + ' loops {0:4.0f} and tolerance is {1:e}'.format(87.0, 4.905623e-06)
    assert line == '{lev: 500} loops   87 and tolerance is 4.905623e-06'


def test_cal_flow_gill_matsuno_matches_reference_formula():
    lat = np.linspace(-90, 90, 19); lon = np.linspace(0, 360, 24)
    rng = np.random.default_rng(0)
    S = Field(rng.standard_normal((19, 24)), ('lat', 'lon'), {'lat': lat, 'lon': lon})
    u, v = apps.cal_flow(S, ['lat', 'lon'], vtype='GillMatsuno', mParams={'epsilon': 1e-5, 'Phi': 5000})
    f = 2 * 7.292e-5 * np.sin(np.deg2rad(lat)); eps = 1e-5
    c1 = (eps / (eps**2 + f**2))[:, None]; c2 = (f / (eps**2 + f**2))[:, None]
    d = np.deg2rad(1.0) * 6371200.0
    Sx = np.gradient(S.values, lon, axis=1); Sy = np.gradient(S.values, lat, axis=0)
    cosL = np.cos(np.deg2rad(lat))[:, None]
    assert np.array_equal(u.values, -c1 * Sx / d / cosL - c2 * Sy / d)
    assert np.array_equal(v.values, -c1 * Sy / d + c2 * Sx / d / cosL)
    with pytest.raises(Exception, match='unsupported vtype'):
        apps.cal_flow(S, ['lat', 'lon'], vtype='vorticity')


def test_cal_flow_streamfunction_and_potential():
    """reference apps.py:1207-1273 + finitediffs.py:548-659: centred differences on the BC-padded
    field.  Solid-body rotation psi = -a*U*sin(lat) gives u = U*cos(lat), v = 0; periodic padding
    makes d/dlon exact across the date line; 'fixed' pads with 0, 'extend' with the edge value."""
    Re = 6371200.0
    lat = np.linspace(-80, 80, 33); lon = np.arange(0, 360, 7.5)
    la = np.deg2rad(lat)[:, None]; lo = np.deg2rad(lon)[None, :]
    psi = Field(-Re * 10.0 * np.sin(la) + 0 * lo, ('lat', 'lon'), {'lat': lat, 'lon': lon})
    u, v = apps.cal_flow(psi, ['lat', 'lon'], BCs=['extend', 'periodic'])
    assert np.allclose(u.values[1:-1], 10.0 * np.cos(la)[1:-1], rtol=2e-3) and np.abs(v.values).max() == 0
    wave = Field(Re * np.cos(la) * np.sin(2 * lo), ('lat', 'lon'), {'lat': lat, 'lon': lon})
    u, v = apps.cal_flow(wave, ['lat', 'lon'], BCs=['extend', 'periodic'])
    assert np.allclose(v.values, 2 * np.cos(2 * lo) * np.ones_like(la), atol=0.03)     # across the date line too
    up, vp = apps.cal_flow(wave, ['lat', 'lon'], BCs=['extend', 'periodic'], vtype='velocitypotential')
    assert np.array_equal(up.values, v.values) and np.array_equal(vp.values, -u.values)
    # boundary padding: first row uses (S[1] - pad) / (2 dy)
    y = np.linspace(0, 4e5, 5); x = np.linspace(0, 6e5, 7)
    Sv = np.random.default_rng(3).standard_normal((5, 7))
    S = Field(Sv, ('y', 'x'), {'y': y, 'x': x})
    u, v = apps.cal_flow(S, ['y', 'x'], coords='cartesian', BCs=['fixed', 'extend'])
    assert np.allclose(-u.values[0], (Sv[1] - 0.0) / 2e5) and np.allclose(-u.values[2], (Sv[3] - Sv[1]) / 2e5)
    assert np.allclose(v.values[:, 0], (Sv[:, 1] - Sv[:, 0]) / 2e5) and np.allclose(v.values[:, -1], (Sv[:, -1] - Sv[:, -2]) / 2e5)
    # vertical planes
    lev = np.linspace(1e5, 2e4, 9)
    Z = Field(np.random.default_rng(4).standard_normal((9, 33)), ('lev', 'lat'), {'lev': lev, 'lat': lat})
    a, b = apps.cal_flow(Z, ['lev', 'lat'], coords='z-lat', BCs=['extend', 'extend'])
    cs = np.cos(np.deg2rad(lat))[None, :]
    assert np.allclose(-a.values[3], (Z.values[4] - Z.values[2]) / (lev[4] - lev[2]) / cs[0])
    assert np.allclose(b.values[:, 5], (Z.values[:, 6] - Z.values[:, 4]) / (np.deg2rad(lat[6] - lat[4]) * Re) / cs[0, 5])


def test_xarray_like_objects_are_accepted_and_returned():
    """The boundary accepts anything with .dims/.coords/.values (xarray.DataArray) -- exercised
    with a stand-in because xarray is not installed in this image."""
    from xinvert_amd.field import from_any, to_like

    class Coord:
        def __init__(self, v): self.values = np.asarray(v)

    class FakeDA:
        def __init__(self, v, dims, coords):
            self.values, self.dims, self.name = np.asarray(v), tuple(dims), 'x'
            self.coords = {k: Coord(c) for k, c in coords.items()}

    da = FakeDA(np.zeros((3, 4)), ('lat', 'lon'), {'lat': [0., 1., 2.], 'lon': [0., 10., 20., 30.]})
    f = from_any(da)
    assert isinstance(f, Field) and f.dims == ('lat', 'lon') and list(f['lon']) == [0., 10., 20., 30.]
    assert to_like(f, f) is f
    assert to_like(f, da) is f            # no xarray here: the Field comes back unchanged


def test_labelled_coefficients_align_by_dim_name():
    """A labelled coefficient or icbc whose dims are ordered differently from the forcing is lined
    up by NAME (xarray semantics, reference apps.py `.loc` arithmetic), never by shape coincidence:
    a square core is the case where a shape test cannot tell."""
    from xinvert_amd import core, apps
    n = 6
    y = np.arange(n, dtype=float); x = np.arange(n, dtype=float) * 2
    F = Field(np.zeros((3, n, n)), ('t', 'y', 'x'), {'y': y, 'x': x})
    perm, _, _ = core._batch_layout(F, ['y', 'x'])
    base = np.arange(n * n, dtype=float).reshape(n, n)             # [y, x]
    a, st, rc = core._prep_coef(Field(base.T.copy(), ('x', 'y')), F, perm, (n, n), 3)
    assert st == 0 and not rc and np.array_equal(a, base)         # transposed back to [y, x]
    a, st, rc = core._prep_coef(Field(base, ('y', 'x')), F, perm, (n, n), 3)
    assert st == 0 and np.array_equal(a, base)
    a, st, rc = core._prep_coef(base, F, perm, (n, n), 3)          # bare ndarray: core layout
    assert st == 0 and np.array_equal(a, base)
    with pytest.raises(Exception):
        core._prep_coef(Field(base, ('y', 'z')), F, perm, (n, n), 3)
    # icbc with swapped dims
    iP = apps._update(apps.default_iParams, {'BCs': ['fixed', 'fixed']})
    ic = Field(base.T.copy(), ('x', 'y'), {'y': y, 'x': x})
    _, initS, _ = apps._mask_FS(F, ['y', 'x'], iP, ic)
    assert np.array_equal(initS.values[1][0], base[0]) and np.array_equal(initS.values[2][:, -1], base[:, -1])
    assert (initS.values[:, 1:-1, 1:-1] == 0).all()


def test_xarray_edge_round_trip_through_a_standin_module():
    """field.from_any / field.to_like: a DataArray-shaped object in, a `xarray.DataArray` out (reference apps.py:1389-1392
    returns a DataArray named 'inverted').  xarray is not installed in the build image: a minimal stand-in module is injected
    so that the return branch executes (VERDICT r5 hygiene item; the solve between the two is GPU work:
    tests/test_gpu_frontend.py::test_invert_poisson_dataarray_in_dataarray_out)."""
    import util
    from xinvert_amd import field
    lat, lon = np.linspace(-60., 60., 7), np.arange(0., 360., 30.)
    vals = np.arange(7. * 12).reshape(7, 12)
    with util.xarray_standin() as xr:
        da = xr.DataArray(vals, dims=('lat', 'lon'), coords={'lat': lat, 'lon': lon}, name='vor')
        F = field.from_any(da)
        assert isinstance(F, field.Field) and F.dims == ('lat', 'lon') and F.name == 'vor'
        assert np.array_equal(F['lat'], lat) and np.array_equal(F['lon'], lon) and np.array_equal(F.values, vals)
        out = field.to_like(F.like(vals * 2.0, 'inverted'), da)
        assert isinstance(out, xr.DataArray) and out.name == 'inverted' and out.dims == ('lat', 'lon')
        assert np.array_equal(out.values, vals * 2.0) and np.array_equal(out.coords['lat'].values, lat)
        # a Field template, or a bare ndarray, comes back as the Field
        f2 = F.like(vals, 'x')
        assert field.to_like(f2, F) is f2 and field.to_like(f2, vals) is f2
