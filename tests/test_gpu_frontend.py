"""GPU: the reference-shaped front end (invert_* -> core.inv_* -> C-ABI -> HIP) on the
reference's own test cases and the committed golden fixtures, plus size-independent properties
at BASELINE.json's full sizes."""
import numpy as np
import pytest

import util
from util import golden, U

pytestmark = pytest.mark.gpu

LEX, AUTO, C2 = 0, 1, 2


def _gm_fields():
    d = golden('gill_matsuno.npz')
    return d, d['lat'], d['lon']


def test_invert_gill_matsuno_known_answers(capsys):
    """reference tests/test_GillMatsuno.py:14-57 through xinvert_amd.invert_GillMatsuno:
    same asserts as the reference test (KE sums with np.isclose), batched over the 3 forcings."""
    import xinvert_amd as xa
    d, lat, lon = _gm_fields()
    Q = xa.Field(np.stack([d['Q1'], d['Q2'], d['Q3']]), ('case', 'lat', 'lon'),
                 {'case': np.arange(3), 'lat': lat, 'lon': lon})
    iParams = {'BCs': ['fixed', 'periodic'], 'mxLoop': 2000, 'tolerance': 1e-8, 'optArg': 1.4}
    mParams = {'epsilon': 1e-5, 'Phi': 5000}
    h = xa.invert_GillMatsuno(Q, dims=['lat', 'lon'], iParams=iParams, mParams=mParams)
    out = capsys.readouterr().out.strip().splitlines()
    assert len(out) == 3 and out[0].startswith('{case: 0} loops ') and 'tolerance is' in out[0]
    u, v = xa.cal_flow(h, dims=['lat', 'lon'], BCs=['fixed', 'periodic'], mParams=mParams, vtype='GillMatsuno')
    ke = ((u.values**2 + v.values**2) / 2).sum(axis=(1, 2))
    assert (h.values[0] <= 0).all() and (np.abs(h.values[1]) <= 370).all() and (h.values[2] <= 0).all()
    assert np.isclose(ke[0], 4351.62244687) and np.isclose(ke[1], 5833.33192343) and np.isclose(ke[2], 5100.85325027)
    fl = h.iParams['flags']
    assert fl.shape == (3, 3) and (fl[:, 0] == 0).all() and (fl[:, 1] < 1e-8).all()
    assert h.iParams['stats']['path'] == 2


def test_gill_matsuno_converged_vs_reference_fields():
    """Converged HIP field vs the converged lexicographic (reference-ordering) oracle: <= 1e-6."""
    from test_oracle_golden import _gm_problem
    d, lat, lon = _gm_fields()
    for name in ('Q1', 'Q2', 'Q3'):
        p, _ = _gm_problem(d[name], lat, lon, 4000, 1e-13)
        Sl, _ = util.run_oracle(p, 4000, 1e-13, LEX)
        Sh, fl, _ = util.run_hip_batched([p], 4000, 1e-13)
        assert util.rel_l2(Sh[0], Sl) < 1e-6


def test_invert_stommel_reference_case():
    """reference tests/test_StommelWBC.py:14-55 (S1, S2): converged streamfunction within 1e-6
    of the stored reference field (S2) and of the survey-time scalar pins (S1)."""
    import xinvert_amd as xa
    d = golden('stommel.npz')
    curl = xa.Field(d['curl'], ('ydef', 'xdef'), {'ydef': d['ydef'], 'xdef': d['xdef']})
    iParams = {'BCs': ['fixed', 'fixed'], 'mxLoop': 5000, 'optArg': 1.9, 'tolerance': 1e-12,
               'printInfo': False}
    S2 = xa.invert_Stommel(curl, dims=['ydef', 'xdef'], coords='cartesian', iParams=iParams,
                           mParams={'beta': 1.8e-11, 'R': 0.0008, 'D': 200})
    assert util.rel_l2(S2.values, d['S2']) < 1e-6
    S1 = xa.invert_Stommel(curl, dims=['ydef', 'xdef'], coords='cartesian', iParams=iParams,
                           mParams={'beta': 0, 'R': 0.0008, 'D': 200})
    assert abs(S1.values.max() / 6.1120365308e+05 - 1) < 1e-6
    assert abs(np.abs(S1.values).mean() / 2.7816492185e+05 - 1) < 1e-6


def test_invert_stommel_munk_reference_case():
    """reference tests/test_MunkWBC.py:14-58 through invert_StommelMunk.  h2 (A4 = 5e2) converges:
    the reference's own assert holds.  h1's pin is an un-converged lexicographic iterate (see
    tests/test_oracle_golden.py::test_munk_known_answers); for it the HIP result is checked bit for
    bit against the oracle's 9-colour ordering at the same loop count."""
    import xinvert_amd as xa
    from test_oracle_golden import munk_problem
    p, curl, x, y = munk_problem(5e2)
    F = xa.Field(curl, ('ydef', 'xdef'), {'ydef': y, 'xdef': x})
    iParams = {'BCs': ['fixed', 'fixed'], 'mxLoop': 4000, 'tolerance': 1e-14, 'optArg': 1.0,
               'undef': np.nan, 'printInfo': False}
    h2 = xa.invert_StommelMunk(F, dims=['ydef', 'xdef'], coords='cartesian', iParams=iParams,
                               mParams={'A4': 5e2, 'beta': 1.8e-11, 'R': 0.0001, 'D': 200})
    assert h2.values.shape == curl.shape and np.isclose(h2.values.max(), 399667.8611556)
    Sl, _ = util.run_oracle(p, 4000, 1e-14, LEX)
    assert util.rel_l2(h2.values, Sl) < 1e-6
    p1, _, _, _ = munk_problem(5e3)
    So, flo = util.run_oracle(p1, 4000, 1e-14, AUTO)
    h1 = xa.invert_StommelMunk(F, dims=['ydef', 'xdef'], coords='cartesian', iParams=iParams,
                               mParams={'A4': 5e3, 'beta': 1.8e-11, 'R': 0.0001, 'D': 200})
    assert np.array_equal(h1.values, So) and h1.iParams['flags'][2] == flo[2]


def test_invert_fofonoff_reference_case():
    """reference tests/test_Fofonoff.py:13-44 through invert_Fofonoff (fused kernel, x-uniform
    A, D, E): shape/dims asserts of the reference + converged field vs the reference ordering."""
    import xinvert_amd as xa
    from test_oracle_golden import fofonoff_problem
    p, Fv, xc, yc = fofonoff_problem()
    F = xa.Field(Fv, ('y', 'x'), {'y': yc, 'x': xc})
    assert F.shape == (251, 301)
    sf = xa.invert_Fofonoff(F, dims=['y', 'x'], coords='cartesian',
                            iParams={'BCs': ['fixed', 'fixed'], 'mxLoop': 2000, 'tolerance': 1e-14,
                                     'optArg': 1.2, 'printInfo': False},
                            mParams={'f0': 1e-4, 'beta': 2e-11, 'c0': 8e-9, 'c1': 1e-4})
    assert sf.dims == F.dims and sf.shape == F.shape
    assert sf.iParams['stats']['path'] == 2 and sf.iParams['stats']['xuniform_mask'] == 7
    Sl, _ = util.run_oracle(p, 4000, 1e-14, LEX)
    assert util.rel_l2(sf.values, Sl) < 1e-6


def test_invert_bretherton_synthetic_topography():
    """invert_BrethertonHaidvogel (reference tests/test_Bretherton.py:13-31 call shape) on a
    synthetic seamount: HIP bit-equal to the oracle's ordering, converged within 1e-6 of the
    reference ordering."""
    import xinvert_amd as xa
    from xinvert_amd import apps
    x = np.linspace(0, 3e5, 121); y = np.linspace(0, 2e5, 81)
    topo = 300.0 * np.exp(-(((x[None, :] - 1.5e5) / 4e4) ** 2 + ((y[:, None] - 1e5) / 3e4) ** 2))
    topo = topo - topo.mean()
    h = xa.Field(topo, ('y', 'x'), {'y': y, 'x': x})
    # fixed sweep count for the bitwise check: a stop test at 1e-14 sits in the rounding noise of
    # the norm, whose summation order differs between the device and the serial oracle
    iParams = {'BCs': ['fixed', 'fixed'], 'mxLoop': 300, 'tolerance': 0.0, 'undef': np.nan, 'printInfo': False}
    mParams = {'f0': 1e-4, 'D': 1000, 'lambda': 1e-15}
    S1 = xa.invert_BrethertonHaidvogel(h, dims=['y', 'x'], coords='cartesian', mParams=mParams, iParams=iParams)
    iP = apps._update(apps.default_iParams, iParams)
    mP = apps._update(apps.default_mParams, mParams)
    Fm, initS, cs = apps._coeffs_Bretherton(h, ['y', 'x'], 'cartesian', mP, iP, None)
    ps = apps._cal_params2D(y, x, 'cartesian')
    p = dict(kind='std2dt', yc=81, xc=121, BCy='fixed', BCx='fixed', dely=ps['del2'], delx=ps['del1'],
             delxSqr=ps['del1Sqr'], ratio=ps['ratio'], ratioQtr=ps['ratioQtr'], ratioSqr=ps['ratioSqr'],
             optArg=ps['optArg'], undef=U, S0=np.zeros((81, 121)),
             coefs=[np.ascontiguousarray(c) for c in cs] + [Fm.values])
    So, flo = util.run_oracle(p, 300, 0.0, C2)
    assert np.array_equal(S1.values, So) and S1.iParams['flags'][2] == flo[2] == 300
    Sl, _ = util.run_oracle(p, 6000, 1e-15, LEX)
    Sh, _, _ = util.run_hip_batched([p], 6000, 1e-15)
    assert util.rel_l2(Sh[0], Sl) < 1e-6


def test_invert_bretherton_reference_ke_pin():
    """reference tests/test_Bretherton.py:13-42 as written: Data/topo.nc, `invert_BrethertonHaidvogel`
    + `cal_flow`, KE = sum(u^2 + v^2) / 2 must satisfy np.isclose(KE, 0.0812731) -- the reference's
    own assert, on the HIP path."""
    import xinvert_amd as xa
    g = util.golden('topo.npz')
    topo = g['topo'] - g['topo'].mean()
    h = xa.Field(topo, ('y', 'x'), {'y': g['y'], 'x': g['x']})
    assert h.dims == ('y', 'x') and h.shape == (201, 301)
    iParams = {'BCs': ['fixed', 'fixed'], 'mxLoop': 3000, 'tolerance': 1e-16, 'undef': np.nan, 'printInfo': False}
    mParams1 = {'f0': 1e-4, 'D': 1000, 'lambda': 1e-15}
    S1 = xa.invert_BrethertonHaidvogel(h, dims=['y', 'x'], coords='cartesian', mParams=mParams1, iParams=iParams)
    u1, v1 = xa.cal_flow(S1, dims=['y', 'x'], coords='cartesian')
    assert S1.dims == h.dims and S1.shape == h.shape and u1.dims == h.dims and u1.shape == h.shape
    KE = float((u1.values ** 2 + v1.values ** 2).sum() / 2)
    assert np.isclose(KE, 0.0812731), KE


def test_invert_ishida_mask_periodic_odd_width():
    """reference tests/test_Ishida.py:13-63 (h1, h2): value-undef mask, periodic x, xc = 251."""
    import xinvert_amd as xa
    xnum, ynum = 251, 151
    Lx, Ly = 1e7, 2 * np.pi * 1e6
    x = np.linspace(0, Lx, xnum); y = np.linspace(0, Ly, ynum)
    curl = -np.pi * np.sin(2. * np.pi * (y[:, None] + 0 * x[None, :]) / Ly) / Ly
    curl[65:, 100:104] = -9999
    curl[:75, 130:134] = -9999
    F = xa.Field(curl, ('ydef', 'xdef'), {'ydef': y, 'xdef': x})
    iParams = {'BCs': ['fixed', 'periodic'], 'mxLoop': 3000, 'tolerance': 1e-9, 'optArg': 1.4,
               'undef': -9999, 'printInfo': False}
    h1 = xa.invert_Stommel(F, dims=['ydef', 'xdef'], coords='cartesian', iParams=iParams,
                           mParams={'beta': 2.2e-11, 'R': 0.0009, 'D': 200})
    h2 = xa.invert_Stommel(F, dims=['ydef', 'xdef'], coords='cartesian', iParams=iParams,
                           mParams={'beta': 2.2e-11, 'R': 0.0009 * 20, 'D': 200})
    land = curl == -9999
    assert (h1.values[land] == -9999).all()                     # de-masked with iParams['undef']
    assert (np.abs(h1.values[~land]) <= 5.5e5).all() and (np.abs(h2.values[~land]) <= 2.8e4).all()
    assert h1.iParams['stats']['colours'] == 4                   # red-black + the odd-width seam


def test_invert_poisson_real_data_both_times():
    """Data/Helmholtz_atmos.nc vorticity (2 x 73 x 144 f32), reference tests/test_Poisson.py:14-24
    parameters; converged HIP vs converged reference-ordering oracle after removing the mean
    (pure-Neumann/periodic problem: SURVEY N10)."""
    import xinvert_amd as xa
    from xinvert_amd import apps
    d = golden('poisson_atmos.npz')
    lat, lon = d['lat'].astype(np.float64), d['lon'].astype(np.float64)
    vor = xa.Field(d['vor_f32'], ('time', 'lat', 'lon'), {'time': np.arange(2), 'lat': lat, 'lon': lon})
    sf = xa.invert_Poisson(vor, dims=['lat', 'lon'],
                           iParams={'BCs': ['fixed', 'periodic'], 'mxLoop': 5000, 'tolerance': 1e-13,
                                    'printInfo': False})
    assert sf.values.shape == (2, 73, 144)
    iP = apps._update(apps.default_iParams, {'BCs': ['fixed', 'periodic']})
    Fv = xa.Field(d['vor_f32'].astype(np.float64), ('time', 'lat', 'lon'), {'lat': lat, 'lon': lon})
    F, initS, (A, B, C) = apps._coeffs_Poisson(Fv, ['lat', 'lon'], 'lat-lon', apps.default_mParams, iP, None)
    ps = apps._cal_params2D(lat, lon, 'lat-lon')
    for t in range(2):
        p = dict(kind='std2d', yc=73, xc=144, BCy='fixed', BCx='periodic', dely=ps['del2'], delx=ps['del1'],
                 delxSqr=ps['del1Sqr'], ratioQtr=ps['ratioQtr'], ratioSqr=ps['ratioSqr'],
                 optArg=ps['optArg'], undef=U, S0=np.zeros((73, 144)),
                 coefs=[A, B, C, np.ascontiguousarray(F.values[t])])
        Sl, _ = util.run_oracle(p, 5000, 1e-13, LEX)
        assert util.rel_l2(sf.values[t], Sl) < 1e-6


def test_reference_poisson_test_assertion_on_real_data():
    """The reference's own check (tests/test_Poisson.py:14-41): invert the bundled vorticity with
    BCs ['extend', 'periodic'], tol 1e-11, take the rotational wind with cal_flow and verify that
    its divergence vanishes away from the poles (FiniteDiff.divg restated: finitediffs.py:208-272).
    Added here: the inverted streamfunction satisfies the discrete Poisson equation it was built
    from (residual of the 5-point operator relative to the forcing)."""
    import xinvert_amd as xa
    from xinvert_amd import apps
    d = golden('poisson_atmos.npz')
    lat, lon = d['lat'].astype(np.float64), d['lon'].astype(np.float64)
    vor = xa.Field(d['vor_f32'], ('time', 'lat', 'lon'), {'time': np.arange(2), 'lat': lat, 'lon': lon})
    iParams = {'BCs': ['extend', 'periodic'], 'undef': np.nan, 'mxLoop': 5000, 'tolerance': 1e-11,
               'printInfo': False}
    sf = xa.invert_Poisson(vor, dims=['lat', 'lon'], iParams=iParams)
    us, vs = xa.cal_flow(sf, dims=['lat', 'lon'], BCs=iParams['BCs'], vtype='streamfunction')
    deg2m = np.pi * 6371200.0 / 180.0
    cos = np.cos(np.deg2rad(lat))[None, :, None]
    div0 = apps._deriv_center(us.values, 2, lon, 'periodic', deg2m * cos) + \
        apps._deriv_center(vs.values * cos, 1, lat, 'extend', deg2m * cos)
    assert np.isclose(div0[:, 1:-1], 0).all()
    assert np.abs(us.values[:, 2:-2]).max() > 5.0               # a real wind field, not zeros
    # discrete residual of the converged field (interior rows): L(psi) == vor * cos(lat) * delx^2 ...
    ps = apps._cal_params2D(lat, lon, 'lat-lon')
    Fv = xa.Field(d['vor_f32'].astype(np.float64), ('time', 'lat', 'lon'), {'lat': lat, 'lon': lon})
    F, _, (A, B, C) = apps._coeffs_Poisson(Fv, ['lat', 'lon'], 'lat-lon', apps.default_mParams,
                                           apps._update(apps.default_iParams, iParams), None)
    for t in range(2):
        S = sf.values[t]
        Se, Sw = np.roll(S, -1, 1), np.roll(S, 1, 1)
        Ce = np.roll(C, -1, 1)
        L = ((A[2:, :] * (S[2:] - S[1:-1]) - A[1:-1] * (S[1:-1] - S[:-2])) * ps['ratioSqr'] +
             (Ce[1:-1] * (Se[1:-1] - S[1:-1]) - C[1:-1] * (S[1:-1] - Sw[1:-1])))
        rhs = F.values[t][1:-1] * ps['del1Sqr']
        assert np.linalg.norm(L - rhs) / np.linalg.norm(rhs) < 1e-5


def test_invert_eliassen_real_data_nine_point():
    """invert_Eliassen on the reference's bundled data (Data/ZonalMean.nc Hadley case,
    tests/test_Eliassen.py:131-147; Data/TC2D.nc, :207-232): the fused 4-colour kernel against (i)
    the REFERENCE's converged field (Hadley: 142 loops) within 1e-6 rel-L2 and (ii) the oracle's
    4-colour ordering bit for bit on the un-converged 600-sweep TC run."""
    import xinvert_amd as xa
    from xinvert_amd import apps
    d = golden('eliassen.npz')
    F = xa.Field(d['hadley_F'], ('LEV', 'lat'), {'LEV': d['hadley_lev'], 'lat': d['hadley_lat']})
    mk = lambda k: xa.Field(d['hadley_' + k], ('LEV', 'lat'), {'LEV': d['hadley_lev'], 'lat': d['hadley_lat']})
    sf = apps.invert_Eliassen(F, dims=['LEV', 'lat'], coords='z-lat',
                              iParams={'BCs': ['fixed', 'fixed'], 'mxLoop': 3000, 'tolerance': 1e-13,
                                       'printInfo': False},
                              mParams={'A': mk('A'), 'B': mk('B'), 'C': mk('C')})
    assert sf.dims == F.dims and sf.shape == F.shape
    ok = ~np.isnan(d['hadley_F'])
    assert util.rel_l2(sf.values[ok], d['hadley_S'][ok]) < 1e-6
    # coefficients handed in with swapped dim order are aligned by name
    sw = lambda k: xa.Field(d['hadley_' + k].T, ('lat', 'LEV'), {'LEV': d['hadley_lev'], 'lat': d['hadley_lat']})
    sf2 = apps.invert_Eliassen(F, dims=['LEV', 'lat'], coords='z-lat',
                               iParams={'BCs': ['fixed', 'fixed'], 'mxLoop': 3000, 'tolerance': 1e-13,
                                        'printInfo': False},
                               mParams={'A': sw('A'), 'B': sw('B'), 'C': sw('C')})
    assert np.array_equal(sf2.values, sf.values, equal_nan=True)
    p, _, _ = util.eliassen_problem('tc')
    So, flo = util.run_oracle(p, 600, 1e-12, AUTO)
    S, fl, st = util.run_hip_batched([p], 600, 1e-12)
    assert st['colours'] == 4 and st['path'] == 2
    assert np.array_equal(S[0], So) and fl[0][2] == flo[2] == 600


_APPS = {
    # name: (entry, coefficient builder, kernel kind, coords, mParams, valid)
    'GillMatsuno_test': ('invert_GillMatsuno_test', '_coeffs_GillMatsuno_test', 'std2dt', 'lat-lon',
                         {'epsilon': 1e-4, 'Phi': 5000.}, ['f0', 'beta', 'epsilon', 'Phi', 'g', 'Omega', 'Rearth']),
    'Stommel_test': ('invert_Stommel_test', '_coeffs_Stommel_test', 'std2dt', 'cartesian',
                     {'beta': 1.8e-11, 'R': 8e-4, 'D': 200.}, ['beta', 'R', 'D', 'rho0', 'g', 'Omega', 'Rearth']),
    'StommelArons': ('invert_StommelArons', '_coeffs_StommelArons', 'gen2d', 'lat-lon',
                     {'epsilon': 1e-4}, ['f0', 'beta', 'epsilon', 'g', 'Omega', 'Rearth']),
    'geostrophic': ('invert_geostrophic', '_coeffs_geostrophic', 'std2d', 'lat-lon', {},
                    ['f0', 'beta', 'Omega', 'g', 'Omega', 'Rearth']),
    'PV2D': ('invert_PV2D', '_coeffs_PV2D', 'std2d', 'z-lat', {'f0': 1e-4, 'N2': 2e-4},
             ['f0', 'beta', 'N2', 'g', 'Omega', 'Rearth']),
}


@pytest.mark.parametrize('app', sorted(_APPS))
def test_remaining_apps_converge_to_the_lexicographic_result(app):
    """the other invert_* entry points that sit on the same kernels.  Each
    runs through the front end on a small masked problem and must land within 1e-6 rel-L2 of
    the reference's lexicographic order run on the same coefficients."""
    import xinvert_amd as xa
    from xinvert_amd import apps
    entry, builder, kind, coords, mP, valid = _APPS[app]
    rng = np.random.default_rng(11)
    yc, xc = 30, 48
    if coords == 'lat-lon':
        y = np.linspace(10, 60, yc); x = np.linspace(0, 352.5, xc); BCs = ['fixed', 'periodic']
    elif coords == 'z-lat':
        y = np.linspace(1e5, 2e4, yc); x = np.linspace(10, 60, xc); BCs = ['fixed', 'fixed']
    else:
        y = np.linspace(0, 3e6, yc); x = np.linspace(0, 5e6, xc); BCs = ['fixed', 'fixed']
    Fv = rng.standard_normal((2, yc, xc)) * 1e-10
    # invert_geostrophic rebuilds its forcing from the RAW input compared against -9.99e8
    # (apps.py:1913), so only that sentinel masks it -- as in the reference
    und = U if app == 'geostrophic' else -9999.0
    Fv[:, 12:16, 20:25] = und
    F = xa.Field(Fv, ('t', 'y', 'x'), {'y': y, 'x': x})
    iP = {'BCs': BCs, 'mxLoop': 6000, 'tolerance': 1e-14, 'printInfo': False, 'undef': und}
    if kind == 'std2dt':
        iP['optArg'] = 1.0      # the flux forms carry large antisymmetric cross terms: plain SOR needs omega <= 1
    S = getattr(apps, entry)(F, ['y', 'x'], coords=coords, mParams=mP, iParams=iP)
    assert S.shape == Fv.shape and (S.values[:, 12:16, 20:25] == und).all()
    iPf = apps._update(apps.default_iParams, iP)
    mPf = apps._update(apps.default_mParams, mP, valid)
    H, initS, cs = getattr(apps, builder)(F, ['y', 'x'], coords, mPf, iPf, None)
    ps = apps._cal_params2D(y, x, coords)
    for m in range(2):
        p = dict(kind=kind, yc=yc, xc=xc, BCy=BCs[0], BCx=BCs[1], dely=ps['del2'], delx=ps['del1'],
                 delxSqr=ps['del1Sqr'], ratio=ps['ratio'], ratioQtr=ps['ratioQtr'], ratioSqr=ps['ratioSqr'],
                 optArg=iP.get('optArg', ps['optArg']), undef=U, S0=np.zeros((yc, xc)),
                 coefs=[np.ascontiguousarray(c[m]) for c in cs] + [np.ascontiguousarray(H.values[m])])
        Sl, fll = util.run_oracle(p, 6000, 1e-14, LEX)
        ok = H.values[m] != U
        assert fll[0] == 0 and fll[2] < 6000, fll
        assert util.rel_l2(S.values[m][ok], Sl[ok]) < 1e-6


def test_invert_refstate_vertical_plane():
    """invert_RefState (apps.py:104-145): PV-dependent C coefficient (varies per batch member),
    Gamma profile aligned by dim name; 'Gamma' has no default."""
    import xinvert_amd as xa
    from xinvert_amd import apps
    rng = np.random.default_rng(5)
    th = np.linspace(300, 400, 21); r = np.linspace(1e5, 2.1e6, 41)
    PV = xa.Field(2e-6 * (1 + 0.3 * rng.random((3, 21, 41))), ('t', 'th', 'r'), {'th': th, 'r': r})
    gam = xa.Field(np.linspace(1e-3, 2e-3, 21), ('th',), {'th': th})
    ic = xa.Field(np.broadcast_to((r**2 * 5e-5)[None, None, :], PV.shape).copy(), PV.dims, PV.coords)
    iP = {'BCs': ['fixed', 'fixed'], 'mxLoop': 4000, 'tolerance': 1e-13, 'printInfo': False}
    with pytest.raises(KeyError):
        apps.invert_RefState(PV, ['th', 'r'], coords='cartesian', iParams=iP, icbc=ic)
    S = apps.invert_RefState(PV, ['th', 'r'], coords='cartesian', mParams={'Gamma': gam}, iParams=iP, icbc=ic)
    assert S.iParams['stats']['path'] == 2
    mPf = apps._update(apps.default_mParams, {'Gamma': gam}, ['Ang0', 'Gamma', 'g', 'Omega', 'Rearth'])
    iPf = apps._update(apps.default_iParams, iP)
    H, initS, cs = apps._coeffs_RefState(PV, ['th', 'r'], 'cartesian', mPf, iPf, ic)
    assert cs[2].strides[0] != 0 and cs[0].strides[0] == 0          # C per member, A shared
    ps = apps._cal_params2D(th, r, 'cartesian')
    for m in range(3):
        p = dict(kind='std2d', yc=21, xc=41, BCy='fixed', BCx='fixed', dely=ps['del2'], delx=ps['del1'],
                 delxSqr=ps['del1Sqr'], ratioQtr=ps['ratioQtr'], ratioSqr=ps['ratioSqr'], optArg=ps['optArg'],
                 undef=U, S0=initS.values[m].copy(),
                 coefs=[np.ascontiguousarray(c[m]) for c in cs] + [np.ascontiguousarray(H.values[m])])
        Sl, fll = util.run_oracle(p, 4000, 1e-13, LEX)
        assert fll[2] < 4000 and util.rel_l2(S.values[m], Sl) < 1e-6


def test_animate_iteration_poisson_real_data():
    """reference tests/test_AnimateConverge.py:13-31: 40 frames of (1+1) sweeps on the bundled
    vorticity; frames must equal single solves of the same total sweep count (restartability)."""
    import xinvert_amd as xa
    d = golden('poisson_atmos.npz')
    lat, lon = d['lat'].astype(np.float64), d['lon'].astype(np.float64)
    vor = xa.Field(d['vor_f32'][0], ('lat', 'lon'), {'lat': lat, 'lon': lon})
    iParams = {'BCs': ['fixed', 'periodic'], 'tolerance': 1e-30}
    sf = xa.animate_iteration('Poisson', vor, dims=['lat', 'lon'], iParams=iParams,
                              loop_per_frame=1, max_frames=40)
    assert sf.dims == ('iter', 'lat', 'lon') and sf.shape == (40, 73, 144)
    assert list(sf['iter'][:3]) == [1, 2, 3]
    one = xa.invert_Poisson(vor, dims=['lat', 'lon'],
                            iParams={'BCs': ['fixed', 'periodic'], 'tolerance': 1e-30, 'mxLoop': 79,
                                     'printInfo': False})
    assert np.array_equal(sf.values[-1], one.values)           # 40 frames x 2 sweeps == 80 sweeps
    # flags and statistics of the frames are left where the inv_* calls leave them: in the caller's iParams
    assert iParams['frame_flags'].shape == (40, 3) and (iParams['frame_flags'][:, 2] == 1).all()
    assert np.array_equal(iParams['flags'], iParams['frame_flags'][-1]) and iParams['stats']['path'] in (1, 2)
    # engine options in iParams reach the resident solves too: the colour path, same bits
    ip2 = {'BCs': ['fixed', 'periodic'], 'tolerance': 1e-30, 'engine_path': 1}
    sf2 = xa.animate_iteration('Poisson', vor, dims=['lat', 'lon'], iParams=ip2, loop_per_frame=1, max_frames=5)
    assert ip2['stats']['path'] == 1 and np.array_equal(sf2.values, sf.values[:5])
    with pytest.raises(Exception, match='unsupported problem'):
        xa.animate_iteration('nonsense', vor, dims=['lat', 'lon'])


@pytest.mark.parametrize('coords', ['lat-lon', 'cartesian'])
def test_gill_matsuno_flow_on_device(coords):
    """xinv_gm_flow_f64_dev == apps.cal_flow(vtype='GillMatsuno') bit for bit (uniform lat axis,
    non-uniform-in-the-last-bit lon axis = both numpy.gradient forms), batched."""
    import torch
    import xinvert_amd as xa
    from xinvert_amd import apps
    rng = np.random.default_rng(4)
    lat = np.linspace(-90, 90, 73) if coords == 'lat-lon' else np.linspace(-2e6, 2e6, 73)
    lon = np.linspace(0, 360, 144) if coords == 'lat-lon' else np.linspace(0, 1e7, 144)
    if coords == 'lat-lon':
        lat = lat[1:-1]                       # keep cos(lat) away from zero for a clean comparison
    phi = rng.standard_normal((3, lat.size, lon.size)) * 100.0
    mP = {'epsilon': 1e-5, 'Phi': 5000}
    u0, v0 = apps.cal_flow(xa.Field(phi, ('m', 'lat', 'lon'), {'lat': lat, 'lon': lon}), ['lat', 'lon'],
                           coords=coords, vtype='GillMatsuno', mParams=mP)
    S = torch.from_numpy(phi).cuda(); u = torch.empty_like(S); v = torch.empty_like(S)
    apps.cal_flow_gm_device(S.data_ptr(), u.data_ptr(), v.data_ptr(), 3, lat, lon, coords=coords, mParams=mP)
    assert np.array_equal(u.cpu().numpy(), u0.values) and np.array_equal(v.cpu().numpy(), v0.values)


def test_invert_omega_3d_small():
    import xinvert_amd as xa
    from xinvert_amd import synthetic
    p = synthetic.omega_latlon(8, 24, 36, 2)
    S, fl, st = util.run_hip_batched([synthetic.member(p, m) for m in range(2)], 3000, 1e-14, shared=p['shared'])
    for m in range(2):
        Sl, _ = util.run_oracle(synthetic.member(p, m), 3000, 1e-14, LEX)
        assert util.rel_l2(S[m], Sl) < 1e-6


@pytest.mark.parametrize('coords', ['lat-lon', 'cartesian'])
def test_invert_3DOcean_matches_lexicographic_oracle(coords):
    """apps.invert_3DOcean (reference apps.py:830-888, 2055-2109) on a small masked volume with a
    stratification profile: converged red-black result within 1e-6 rel-L2 of the reference's
    lexicographic order run on the same coefficients."""
    import xinvert_amd as xa
    from xinvert_amd import apps
    rng = np.random.default_rng(7)
    zc, yc, xc = 9, 20, 36
    lev = np.linspace(0., 400., zc)
    lat = np.linspace(-40, 40, yc) if coords == 'lat-lon' else np.linspace(-2e6, 2e6, yc)
    lon = np.linspace(0, 350, xc) if coords == 'lat-lon' else np.linspace(0, 6e6, xc)
    Fv = rng.standard_normal((2, zc, yc, xc)) * 1e-9
    Fv[:, :, 8:12, 10:14] = np.nan
    N2 = xa.Field(np.linspace(2e-4, 1e-4, zc), ('lev',), {'lev': lev})
    F = xa.Field(Fv, ('t', 'lev', 'lat', 'lon'), {'lev': lev, 'lat': lat, 'lon': lon})
    BCs = ['fixed', 'fixed', 'periodic' if coords == 'lat-lon' else 'fixed']
    iP = {'BCs': BCs, 'mxLoop': 4000, 'tolerance': 1e-13, 'printInfo': False}
    mP = {'epsilon': 1e-5, 'N2': N2, 'k': 1e-7, 'f0': 3e-5, 'beta': 2e-11}
    S = apps.invert_3DOcean(F, ['lev', 'lat', 'lon'], coords=coords, mParams=mP, iParams=iP)
    assert S.shape == Fv.shape and np.isnan(S.values[:, :, 8:12, 10:14]).all()
    assert S.iParams['stats']['path'] == 2 and S.iParams['stats']['xuniform_mask'] == 0x7f   # streaming kernel
    # the same coefficients through the oracle's lexicographic order
    mPf = apps._update(apps.default_mParams, mP, ['f0', 'beta', 'epsilon', 'N2', 'k', 'g', 'Omega', 'Rearth'])
    iPf = apps._update(apps.default_iParams, iP)
    H, initS, cs = apps._coeffs_3DOcean(F, ['lev', 'lat', 'lon'], coords, mPf, iPf, None)
    ps = apps._cal_params3D(lev, lat, lon, coords)
    for m in range(2):
        p = dict(kind='gen3d', zc=zc, yc=yc, xc=xc, BCz='fixed', BCy=BCs[1], BCx=BCs[2], delz=ps['del3'],
                 dely=ps['del2'], delx=ps['del1'], delxSqr=ps['del1Sqr'], ratio2=ps['ratio2'],
                 ratio1=ps['ratio1'], ratio2Sqr=ps['ratio2Sqr'], ratio1Sqr=ps['ratio1Sqr'],
                 optArg=ps['optArg'], undef=U, S0=np.zeros((zc, yc, xc)),
                 coefs=[np.ascontiguousarray(c) for c in cs] + [np.ascontiguousarray(H.values[m])])
        Sl, fll = util.run_oracle(p, 4000, 1e-13, LEX)
        ok = H.values[m] != U
        assert fll[2] < 4000 and util.rel_l2(S.values[m][ok], Sl[ok]) < 1e-6


def test_non_trailing_core_dims_and_inplace():
    """Core dims anywhere in F (the reference's .loc[sel].values views): result lands in place."""
    import xinvert_amd as xa
    from xinvert_amd import core, apps
    rng = np.random.default_rng(0)
    y = np.linspace(0, 1e6, 20); x = np.linspace(0, 2e6, 34)
    G = rng.standard_normal((20, 3, 34)) * 1e-10
    F = xa.Field(G, ('y', 'mem', 'x'), {'y': y, 'x': x})
    S = xa.Field(np.zeros_like(G), ('y', 'mem', 'x'), {'y': y, 'x': x})
    ps = apps._cal_params2D(y, x, 'cartesian')
    iP = apps._update(apps._update(apps.default_iParams, {'BCs': ['fixed', 'fixed'], 'mxLoop': 30,
                                                         'tolerance': 0.0, 'printInfo': False}), {})
    iP = apps._update(ps, iP)
    one = np.ones((20, 34)); zero = np.zeros((20, 34))
    buf = S.values
    out = core.inv_general2D(one, zero, one, zero, zero, zero, F, S, ['y', 'x'], iP)
    assert out is S and S.values is buf and np.abs(buf).max() > 0
    for m in range(3):
        p = dict(kind='gen2d', yc=20, xc=34, BCy='fixed', BCx='fixed', dely=ps['del2'], delx=ps['del1'],
                 delxSqr=ps['del1Sqr'], ratio=ps['ratio'], ratioQtr=ps['ratioQtr'], ratioSqr=ps['ratioSqr'],
                 optArg=ps['optArg'], undef=U, S0=np.zeros((20, 34)),
                 coefs=[one, zero, one, zero, zero, zero, np.ascontiguousarray(G[:, m, :])])
        So, flo = util.run_oracle(p, 30, 0.0, C2)
        assert np.array_equal(buf[:, m, :], So)


# ------------------------------------------------------------------ full-size properties
def _full_c2():
    from xinvert_amd import synthetic
    return synthetic.poisson_latlon(1800, 3600, mask=True)


def test_full_size_c2_properties():
    """BASELINE configs[1] (3600 x 1800, land mask).  The oracle cannot finish this size in test
    time, so check size-independent properties: (i) K = 1, 2, 3, 4 fused launches == colour
    path, bit for bit; (ii) land points never change; (iii) the device norm equals a host
    recomputation; (iv) restart: 10 + 10 sweeps == 20 sweeps."""
    from xinvert_amd import synthetic
    p = _full_c2()
    q = synthetic.member(p, 0)
    S1, f1, s1 = util.run_hip_dev([q], 19, 0.0, sweeps_per_launch=1)
    S2, f2, s2 = util.run_hip_dev([q], 19, 0.0, sweeps_per_launch=2)
    S3, f3, s3 = util.run_hip_dev([q], 19, 0.0, path=1)
    assert s1['path'] == 2 and s2['sweeps_per_launch'] == 2 and s3['path'] == 1
    assert s2['masked_tile_pct'] >= 15                   # land blobs + polar caps: skipped tiles
    S4, f4, s4 = util.run_hip_dev([q], 19, 0.0, sweeps_per_launch=2, no_tile_skip=1)
    assert s4['masked_tile_pct'] == 0 and np.array_equal(S2, S4) and np.allclose(f2, f4, rtol=1e-9, atol=1e-12)
    assert np.array_equal(S1, S2) and np.array_equal(S1, S3)
    assert np.allclose(f1, f2, rtol=1e-9, atol=1e-12) and np.allclose(f1, f3, rtol=1e-9, atol=1e-12)
    for spl in (3, 4, 0):                                # deeper passes (20 sweeps = 5x4 = 6x3 + 2) and the default
        S5, f5, s5 = util.run_hip_dev([q], 19, 0.0, sweeps_per_launch=spl)
        assert s5['sweeps_per_launch'] == (spl or 4) and s5['masked_tile_pct'] >= 15
        assert np.array_equal(S1, S5) and np.allclose(f1, f5, rtol=1e-9, atol=1e-12)
    land = q['coefs'][3] == U
    assert land.mean() > 0.2 and (S1[0][land] == 0).all()
    assert np.abs(S1[0][~land]).max() > 0
    Sa, _, _ = util.run_hip_dev([q], 9, 0.0)
    r = dict(q); r['S0'] = Sa[0]
    Sb, _, _ = util.run_hip_dev([r], 9, 0.0)
    assert np.array_equal(Sb, S1)


def test_full_size_c2_sample_rows_against_oracle():
    """A 64-row band of the full problem with frozen ('fixed') band edges is itself a valid SOR
    problem: the HIP result on the band must equal the oracle's, bit for bit."""
    from xinvert_amd import synthetic
    p = _full_c2()
    q = synthetic.member(p, 0)
    j0, j1 = 700, 764
    band = dict(q)
    band['yc'] = j1 - j0
    band['S0'] = np.ascontiguousarray(q['S0'][j0:j1])
    band['coefs'] = [np.ascontiguousarray(c[j0:j1]) for c in q['coefs']]
    So, flo = util.run_oracle(band, 11, 0.0, C2)
    Sh, fl, _ = util.run_hip_batched([band], 11, 0.0, sweeps_per_launch=2)
    assert np.array_equal(Sh[0], So) and fl[0][2] == flo[2]


def test_full_size_c3_stommel_bitwise_paths():
    from xinvert_amd import synthetic
    p = synthetic.stommel_cartesian(2000, 2000)
    q = synthetic.member(p, 0)
    S1, f1, _ = util.run_hip_dev([q], 9, 0.0, sweeps_per_launch=2)
    S2, f2, _ = util.run_hip_dev([q], 9, 0.0, path=1)
    assert np.array_equal(S1, S2)
    assert (S1[0][0] == 0).all() and (S1[0][:, 0] == 0).all() and (S1[0][-1] == 0).all()


def test_full_size_c4_gill_matsuno_members():
    """BASELINE configs[3] at full grid size (720 x 1440), 8 of the 64 members: fused K=2 with every
    latitude-only coefficient read as a per-row scalar == the same kernel streaming every array ==
    the colour path, bit for bit; members are independent (member 3 alone == member 3 in the batch)."""
    from xinvert_amd import synthetic
    p = synthetic.gill_matsuno(720, 1440, 8)
    qs = [synthetic.member(p, m) for m in range(8)]
    S1, f1, s1 = util.run_hip_dev(qs, 11, 0.0, shared=p['shared'])
    S2, f2, s2 = util.run_hip_dev(qs, 11, 0.0, shared=p['shared'], no_xuniform=1)
    S3, f3, s3 = util.run_hip_dev(qs, 11, 0.0, shared=p['shared'], path=1)
    assert s1['path'] == 2 and s1['xuniform_mask'] == 31 and s2['xuniform_mask'] == 0 and s3['path'] == 1
    assert np.array_equal(S1, S2) and np.array_equal(S1, S3)
    Sm, fm, _ = util.run_hip_dev([qs[3]], 11, 0.0)
    assert np.array_equal(Sm[0], S1[3]) and fm[0][2] == f1[3][2]
    assert np.abs(S1).max() > 0 and np.isfinite(S1).all()


def test_full_size_c5_omega_volume():
    """BASELINE configs[4] volume size (50 x 360 x 720, topography mask): the fused 3-D kernel
    (x-uniform coefficients) == the same kernel with every array streamed == the colour path."""
    from xinvert_amd import synthetic
    p = synthetic.omega_latlon(50, 360, 720, 1)
    q = synthetic.member(p, 0)
    S1, f1, s1 = util.run_hip_dev([q], 5, 0.0)
    S2, f2, s2 = util.run_hip_dev([q], 5, 0.0, no_xuniform=1)
    S3, f3, s3 = util.run_hip_dev([q], 5, 0.0, path=1)
    assert s1['path'] == 2 and s1['xuniform_mask'] == 7 and s2['xuniform_mask'] == 0 and s3['path'] == 1
    assert np.array_equal(S1, S2) and np.array_equal(S1, S3)
    below = q['coefs'][3] == U
    assert below.mean() > 0.01 and (S1[0][below] == q['S0'][below]).all()


def test_full_size_c3_munk_three_launch_schemes():
    """BASELINE configs[2], Munk branch, 2000 x 2000: the one-pass kernel, the row-class kernel
    (x-uniform detection off) and nothing-fused must agree bit for bit -- twelve strips, seventy-odd
    row blocks: the size at which a cross-strip race would show."""
    from xinvert_amd import synthetic
    p = synthetic.munk_cartesian(2000, 2000)
    q = synthetic.member(p, 0)
    S1, f1, s1 = util.run_hip_dev([q], 5, 0.0)
    S2, f2, s2 = util.run_hip_dev([q], 5, 0.0, no_xuniform=1)
    S3, f3, s3 = util.run_hip_dev([q], 5, 0.0, path=1)
    assert s1['path'] == 2 and s2['path'] == 1 and s2['xuniform_mask'] == 0 and s3['path'] == 1
    assert np.array_equal(S1, S2) and np.array_equal(S1, S3)
    assert np.allclose(f1, f2, rtol=1e-9, atol=1e-12)
    # a periodic variant (xc % 3 == 0 keeps the fast kernels; the east columns use the stale index)
    r = dict(q); r['BCx'] = 'periodic'; r['xc'] = 1998
    r['S0'] = np.ascontiguousarray(q['S0'][:, :1998]); r['coefs'] = [np.ascontiguousarray(c[:, :1998]) for c in q['coefs']]
    P1, _, t1 = util.run_hip_dev([r], 5, 0.0)
    P2, _, t2 = util.run_hip_dev([r], 5, 0.0, no_xuniform=1)
    assert t1['path'] == 2 and t2['path'] == 1 and np.array_equal(P1, P2)


def test_full_size_nine_point_and_general_3d_paths():
    """2000 x 2000 9-point forms (fused 4-colour kernel vs colour launches) and a 50 x 360 x 720
    general 3-D volume (k_fused3dg vs colour launches): bit for bit."""
    from xinvert_amd import synthetic
    for kind in ('std2d', 'gen2d'):
        q = util.rand2d(kind, 2000, 2000, 'fixed', 'fixed', 1, 1, seed=77)
        S1, f1, s1 = util.run_hip_dev([q], 5, 0.0)
        S2, f2, s2 = util.run_hip_dev([q], 5, 0.0, path=1)
        assert s1['path'] == 2 and s1['colours'] == 4 and s2['path'] == 1 and np.array_equal(S1, S2)
    p = synthetic.ocean3d_latlon(50, 360, 720, 1)
    q = synthetic.member(p, 0)
    S1, f1, s1 = util.run_hip_dev([q], 5, 0.0)
    S2, f2, s2 = util.run_hip_dev([q], 5, 0.0, path=1)
    assert s1['path'] == 2 and s1['xuniform_mask'] == 0x7f and s2['path'] == 1 and np.array_equal(S1, S2)


def test_larger_than_baseline_grid():
    """Four times the BASELINE grid (7200 x 3600, 26 M points, land mask): tall tiles, many strips,
    masked-tile skipping re-planned -- K = 2 with skipping == K = 1 without, bit for bit."""
    from xinvert_amd import synthetic
    p = synthetic.poisson_latlon(3600, 7200, mask=True)
    q = synthetic.member(p, 0)
    S2, f2, s2 = util.run_hip_dev([q], 9, 0.0, sweeps_per_launch=2)
    S1, f1, s1 = util.run_hip_dev([q], 9, 0.0, sweeps_per_launch=1, no_tile_skip=1)
    assert s2['masked_tile_pct'] >= 15 and s1['masked_tile_pct'] == 0
    assert np.array_equal(S1, S2) and np.allclose(f1, f2, rtol=1e-9, atol=1e-12)


def test_integration_md_ctypes_stub_runs():
    """The ctypes stub INTEGRATION.md tells a maintainer of the reference to add (xinvert/hipbind.py)
    is executed as written -- only the library path is filled in -- and must reproduce the oracle."""
    import os
    import re
    from xinvert_amd import _lib
    _lib.require_gpu()                                   # torch first, then the library (single HIP runtime)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    md = open(os.path.join(root, 'INTEGRATION.md')).read()
    code = re.search(r"```python\n(# xinvert/hipbind.py.*?)```", md, re.S).group(1)
    code = code.replace("ctypes.CDLL('libxinv_hip.so')", "ctypes.CDLL(%r)" % _lib.SO)
    ns = {}
    exec(compile(code, 'hipbind.py', 'exec'), ns)
    for kind, fn in (('std2d', 'invert_standard_2D'), ('gen2d', 'invert_general_2D'), ('std3d', 'invert_standard_3D')):
        q = util.rand3d(6, 12, 20, 'fixed', 'periodic', 1, seed=9) if kind == 'std3d' else \
            util.rand2d(kind, 24, 40, 'extend', 'periodic', 0, 1, seed=9)
        So, flo = util.run_oracle(q, 15, 1e-9, AUTO)
        S = q['S0'].copy(); fl = np.array([0., 1., 0.])
        c = [a.copy() for a in q['coefs']]
        if kind == 'std2d':
            ns[fn](S, *c, q['yc'], q['xc'], q['dely'], q['delx'], q['BCy'], q['BCx'], q['delxSqr'],
                   q['ratioQtr'], q['ratioSqr'], q['optArg'], q['undef'], fl, 15, 1e-9)
        elif kind == 'gen2d':
            ns[fn](S, *c, q['yc'], q['xc'], q['dely'], q['delx'], q['BCy'], q['BCx'], q['delxSqr'], q['ratio'],
                   q['ratioQtr'], q['ratioSqr'], q['optArg'], q['undef'], fl, 15, 1e-9)
        else:
            ns[fn](S, *c, q['zc'], q['yc'], q['xc'], q['delz'], q['dely'], q['delx'], q['BCz'], q['BCy'], q['BCx'],
                   q['delxSqr'], q['ratio2Sqr'], q['ratio1Sqr'], q['optArg'], q['undef'], fl, 15, 1e-9)
        assert np.array_equal(S, So) and fl[2] == flo[2], kind


def test_converged_half_degree_poisson_vs_reference_ordering():
    """north_star's acceptance criterion at a mid size: invert_Poisson on a 0.5-degree masked lat-lon
    grid (360 x 720, fixed / periodic) run to convergence on the GPU (red-black, masked tiles skipped)
    against the reference's lexicographic ordering (oracle) run to convergence: rel-L2 <= 1e-6."""
    from xinvert_amd import synthetic
    p = synthetic.poisson_latlon(360, 720, mask=True)
    q = synthetic.member(p, 0)
    S, fl, st = util.run_hip_dev([q], 20000, 1e-13)
    Sl, fll = util.run_oracle(q, 20000, 1e-13, LEX)
    assert fl[0][2] < 20000 and fll[2] < 20000 and st['path'] == 2
    ok = q['coefs'][3] != U
    assert util.rel_l2(S[0][ok], Sl[ok]) < 1e-6
    assert (S[0][~ok] == 0).all()


def test_abs_norm_dev():
    import ctypes
    import torch
    import oracle as orc
    from xinvert_amd import _lib
    L = _lib.require_gpu()
    rng = np.random.default_rng(3)
    a = rng.standard_normal((300, 401)); a[rng.random(a.shape) < 0.2] = U
    t = torch.from_numpy(a).cuda()
    out = ctypes.c_double(0)
    rc = L.xinv_abs_norm_f64_dev(ctypes.c_void_p(t.data_ptr()), a.size, U, ctypes.byref(out), None)
    _lib.check(rc)
    assert abs(out.value / orc.abs_norm(a, U) - 1) < 1e-13


# ------------------------------------------------------------------ front-end passes on the device
def _same(a, b):
    return np.array_equal(a, b, equal_nan=True)


@pytest.mark.parametrize('undef', [np.nan, -9999.0])
def test_device_side_mask_scale_demask_equals_host_front_end(undef):
    """apps.__mask_FS, the builders' F * cos(lat) + re-mask and the output de-mask (reference
    apps.py:2112-2159, 1409-1411, 1389-1392) done by the host entry on the device
    (xinv_options.prep_flags) must give exactly what the numpy front end gives: lat-lon Poisson
    (NaN and value masks, a time axis, chunked members), Gill-Matsuno, 3-D omega."""
    import xinvert_amd as xa
    rng = np.random.default_rng(11)
    lat = np.linspace(-88.75, 88.75, 72); lon = np.arange(0, 360, 2.5)
    la, lo = np.deg2rad(lat)[:, None], np.deg2rad(lon)[None, :]
    vor = np.stack([1e-5 * (np.sin(3 * lo + t) * np.cos(2 * la) + 0.1 * rng.standard_normal((72, 144))) for t in range(5)])
    land = (np.sin(4 * lo + 2 * la) > 0.5) & (np.abs(la) < 1.2)
    vor[:, land] = undef
    vor[3, 10:20, 30:50] = undef                                   # a member-specific mask
    F = xa.Field(vor, ('time', 'lat', 'lon'), {'lat': lat, 'lon': lon})
    for BCs in (['fixed', 'periodic'], ['extend', 'periodic']):
        iP = {'BCs': BCs, 'mxLoop': 60, 'tolerance': 0.0, 'printInfo': False, 'undef': undef, 'host_chunk': 2}
        Sd = xa.invert_Poisson(F, ['lat', 'lon'], iParams=iP)
        Sh = xa.invert_Poisson(F, ['lat', 'lon'], iParams=dict(iP, device_prep=False))
        assert Sd.iParams['stats']['host_chunks'] == 3
        assert _same(Sd.values, Sh.values) and np.array_equal(Sd.iParams['flags'], Sh.iParams['flags'])
        msk = np.isnan(vor) if np.isnan(undef) else (vor == undef)
        assert _same(Sd.values[msk], np.full(msk.sum(), undef)) and np.isfinite(Sd.values[~msk]).all()
        assert np.abs(Sd.values[~msk]).max() > 0
    # cartesian Poisson: no scale
    y = np.linspace(0, 7.1e6, 72); x = np.linspace(0, 1.43e7, 144)
    Fc = xa.Field(vor[1], ('y', 'x'), {'y': y, 'x': x})
    iP = {'BCs': ['fixed', 'fixed'], 'mxLoop': 40, 'tolerance': 0.0, 'printInfo': False, 'undef': undef}
    assert _same(xa.invert_Poisson(Fc, ['y', 'x'], coords='cartesian', iParams=iP).values,
                 xa.invert_Poisson(Fc, ['y', 'x'], coords='cartesian', iParams=dict(iP, device_prep=False)).values)
    # Gill-Matsuno (forcing = Q, no scale) on the same grid
    Q = xa.Field(np.where(np.isnan(vor) | (vor == undef), undef, 0.05 * np.exp(-(la * 6) ** 2 - ((lo - 3) * 3) ** 2)),
                 ('time', 'lat', 'lon'), {'lat': lat, 'lon': lon})
    iP = {'BCs': ['fixed', 'periodic'], 'mxLoop': 50, 'tolerance': 0.0, 'optArg': 1.4, 'printInfo': False, 'undef': undef}
    mP = {'epsilon': 1e-5, 'Phi': 5000}
    assert _same(xa.invert_GillMatsuno(Q, ['lat', 'lon'], mParams=mP, iParams=iP).values,
                 xa.invert_GillMatsuno(Q, ['lat', 'lon'], mParams=mP, iParams=dict(iP, device_prep=False)).values)
    # omega: 3-D, the scale runs along the middle core dim
    lev = np.linspace(1e5, 1e4, 9)
    frc = 1e-17 * rng.standard_normal((2, 9, 72, 144))
    frc[:, :3, land] = undef
    W = xa.Field(frc, ('time', 'lev', 'lat', 'lon'), {'lev': lev, 'lat': lat, 'lon': lon})
    iP = {'BCs': ['fixed', 'fixed', 'periodic'], 'mxLoop': 20, 'tolerance': 0.0, 'printInfo': False, 'undef': undef}
    mP = {'N2': 2e-6}
    Wd = xa.invert_omega(W, ['lev', 'lat', 'lon'], mParams=mP, iParams=iP)
    Wh = xa.invert_omega(W, ['lev', 'lat', 'lon'], mParams=mP, iParams=dict(iP, device_prep=False))
    assert _same(Wd.values, Wh.values) and np.abs(Wd.values[np.isfinite(Wd.values)]).max() > 0
    # ... and as a ROLLING batch (round 6, xinv_hostptr.h: one launch chain over the volumes that have arrived; taken for
    # large batches, here on request): six time steps, an odd sweep budget, float64 and float32 forcing -- the fields of the
    # chunked pipeline, with the front-end passes (mask, cos(lat) scale, zero first guess, de-mask) on the device
    frc6 = 1e-17 * rng.standard_normal((6, 9, 72, 144))
    frc6[:, :3, land] = undef
    frc6[4, 5:, 20:30, 40:60] = undef
    for dt in (np.float64, np.float32):
        W6 = xa.Field(frc6.astype(dt), ('time', 'lev', 'lat', 'lon'), {'lev': lev, 'lat': lat, 'lon': lon})
        iP6 = dict(iP, mxLoop=22)
        Wr = xa.invert_omega(W6, ['lev', 'lat', 'lon'], mParams=mP, iParams=dict(iP6, host_inflight=-1))
        assert Wr.iParams['stats']['host_chunks'] == 6, Wr.iParams['stats']
        Wc = xa.invert_omega(W6, ['lev', 'lat', 'lon'], mParams=mP, iParams=dict(iP6, host_chunk=2))
        assert Wc.iParams['stats']['host_chunks'] == 3
        assert _same(Wr.values, Wc.values) and np.array_equal(Wr.iParams['flags'][:, [0, 2]], Wc.iParams['flags'][:, [0, 2]])


def test_invert_poisson_dataarray_in_dataarray_out():
    """The front end with a DataArray-shaped forcing (a stand-in `xarray` module: the image has none): the result comes back
    as `xarray.DataArray` named 'inverted' with the forcing's dims and coordinates (reference apps.py:1389-1392), and holds
    the numbers the Field call gives."""
    import xinvert_amd as xa
    lat = np.linspace(-87.5, 87.5, 36); lon = np.arange(0., 360., 5.)
    rng = np.random.default_rng(7)
    vor = 1e-5 * rng.standard_normal((36, 72))
    vor[5:9, 10:20] = np.nan
    iP = {'BCs': ['fixed', 'periodic'], 'mxLoop': 60, 'tolerance': 1e-9, 'printInfo': False}
    ref = xa.invert_Poisson(xa.Field(vor, ('lat', 'lon'), {'lat': lat, 'lon': lon}), ['lat', 'lon'], iParams=iP)
    with util.xarray_standin() as xr:
        da = xr.DataArray(vor, dims=('lat', 'lon'), coords={'lat': lat, 'lon': lon}, name='vor')
        out = xa.invert_Poisson(da, ['lat', 'lon'], iParams=iP)
        assert isinstance(out, xr.DataArray) and out.name == 'inverted' and out.dims == ('lat', 'lon')
        assert np.array_equal(out.coords['lon'].values, lon)
        assert np.array_equal(out.values, ref.values, equal_nan=True)
