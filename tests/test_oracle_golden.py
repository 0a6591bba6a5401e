"""The C oracle (lexicographic ordering) against outputs of the reference itself.

Fixtures in tests/golden/ were produced by tests/golden/gen_golden.py running the reference's
own xinvert/numbas.py (pure-Python import) in the build container; the pins quoted from the
reference's tests / executed notebooks are cited per test.  Bar: bit-exact S and flags.
"""
import ast

import numpy as np
import pytest

import util
from util import golden, U

LEX = 0


def _cases():
    d = golden('small_cases.npz')
    return d, [ast.literal_eval(str(m)) for m in d['meta']]


def test_small_cases_bitwise(oracle):
    d, metas = _cases()
    assert len(metas) == 160
    seen = set()
    for m in metas:
        k, kind = m[0], m[1]
        arr = d[k + '_in']
        S = np.ascontiguousarray(arr[0]).copy()
        fl = np.array([0., 1., 0.])
        c = [np.ascontiguousarray(a) for a in arr[1:]]
        if kind == 'std2d':
            _, _, yc, xc, BCy, BCx, dely, delx, om, nsw, tol = m
            r = delx / dely
            oracle.standard_2d(S, *c, yc, xc, dely, delx, BCy, BCx, delx**2, r / 4, r**2, om, U, fl, nsw, tol, LEX)
        elif kind == 'gen2d':
            _, _, yc, xc, BCy, BCx, dely, delx, om, nsw, tol = m
            r = delx / dely
            oracle.general_2d(S, *c, yc, xc, dely, delx, BCy, BCx, delx**2, r, r / 4, r**2, om, U, fl, nsw, tol, LEX)
        else:
            _, _, zc, yc, xc, BCy, BCx, delz, dely, delx, om, nsw, tol = m
            oracle.standard_3d(S, *c, zc, yc, xc, delz, dely, delx, 'fixed', BCy, BCx, delx**2,
                               (delx / delz)**2, (delx / dely)**2, om, U, fl, nsw, tol, LEX)
        assert np.array_equal(S, d[k + '_S']), m
        assert np.array_equal(fl, d[k + '_flags']), m
        seen.add((kind, m[-7] if kind != 'std3d' else m[5], m[-6] if kind != 'std3d' else m[6]))
    # every (kernel x BCy x BCx) cell is present
    assert len(seen) == 2 * 6 + 4


def _gm_problem(Q, lat, lon, mxLoop, tol):
    from xinvert_amd import apps
    from xinvert_amd.field import Field
    iP = apps._update(apps.default_iParams, {'BCs': ['fixed', 'periodic'], 'mxLoop': mxLoop,
                                            'tolerance': tol, 'optArg': 1.4})
    mP = apps._update(apps.default_mParams, {'epsilon': 1e-5, 'Phi': 5000})
    F = Field(Q, ('lat', 'lon'), {'lat': lat, 'lon': lon})
    G, initS, cs = apps._coeffs_GillMatsuno(F, ['lat', 'lon'], 'lat-lon', mP, iP, None)
    ps = apps._cal_params2D(lat, lon, 'lat-lon')
    return dict(kind='gen2d', yc=73, xc=144, BCy='fixed', BCx='periodic', dely=ps['del2'], delx=ps['del1'],
                delxSqr=ps['del1Sqr'], ratio=ps['ratio'], ratioQtr=ps['ratioQtr'], ratioSqr=ps['ratioSqr'],
                optArg=1.4, undef=U, S0=np.zeros((73, 144)), coefs=list(cs) + [G.values]), mP


# printed by the reference in docs/source/notebooks/07_Gill_Matsuno_model.ipynb:103-105
GM_NOTEBOOK = {'Q1': ' 600 and tolerance is 5.608964e-05', 'Q2': '  87 and tolerance is 4.905623e-06',
               'Q3': ' 600 and tolerance is 5.174635e-05'}


@pytest.mark.parametrize('name', ['Q1', 'Q2', 'Q3'])
def test_gill_matsuno_notebook_fields(oracle, name):
    d = golden('gill_matsuno.npz')
    p, _ = _gm_problem(d[name], d['lat'], d['lon'], 600, 1e-5)
    S, fl = util.run_oracle(p, 600, 1e-5, LEX)
    assert np.array_equal(S, d[name + '_S'])
    assert np.array_equal(fl, d[name + '_flags'])
    assert '{0:4.0f} and tolerance is {1:e}'.format(fl[2], fl[1]) == GM_NOTEBOOK[name]


# reference tests/test_GillMatsuno.py:55-57 (np.isclose default rtol 1e-5) and SURVEY 4.3 loops
GM_KE = {'Q1': (4351.62244687, 1628), 'Q2': (5833.33192343, 1146), 'Q3': (5100.85325027, 1618)}


@pytest.mark.parametrize('name', ['Q1', 'Q2', 'Q3'])
def test_gill_matsuno_known_answer_ke(oracle, name):
    from xinvert_amd import apps
    from xinvert_amd.field import Field
    d = golden('gill_matsuno.npz')
    p, mP = _gm_problem(d[name], d['lat'], d['lon'], 2000, 1e-8)
    S, fl = util.run_oracle(p, 2000, 1e-8, LEX)
    assert fl[2] == GM_KE[name][1]
    h = Field(S, ('lat', 'lon'), {'lat': d['lat'], 'lon': d['lon']})
    u, v = apps.cal_flow(h, ['lat', 'lon'], BCs=['fixed', 'periodic'], vtype='GillMatsuno',
                         mParams={'epsilon': 1e-5, 'Phi': 5000})
    ke = ((u.values**2 + v.values**2) / 2).sum()
    assert np.isclose(ke, GM_KE[name][0])
    assert abs(ke / GM_KE[name][0] - 1) < 1e-10
    if name != 'Q2':
        assert (S <= 0).all()
    else:
        assert (np.abs(S) <= 370).all()


def _stommel_problem(beta):
    from xinvert_amd import apps
    from xinvert_amd.field import Field
    d = golden('stommel.npz')
    iP = apps._update(apps.default_iParams, {'BCs': ['fixed', 'fixed'], 'optArg': 1.9})
    mP = apps._update(apps.default_mParams, {'beta': beta, 'R': 0.0008, 'D': 200})
    F = Field(d['curl'], ('ydef', 'xdef'), {'ydef': d['ydef'], 'xdef': d['xdef']})
    G, initS, cs = apps._coeffs_Stommel(F, ['ydef', 'xdef'], 'cartesian', mP, iP, None)
    ps = apps._cal_params2D(d['ydef'], d['xdef'], 'cartesian')
    assert abs(ps['optArg'] - 1.964078724984458) < 1e-14          # SURVEY a8 golden omega
    return dict(kind='gen2d', yc=151, xc=201, BCy='fixed', BCx='fixed', dely=ps['del2'], delx=ps['del1'],
                delxSqr=ps['del1Sqr'], ratio=ps['ratio'], ratioQtr=ps['ratioQtr'], ratioSqr=ps['ratioSqr'],
                optArg=1.9, undef=U, S0=np.zeros((151, 201)), coefs=list(cs) + [G.values]), d


def test_stommel_s2_field(oracle):
    """reference tests/test_StommelWBC.py:14-55, case S2."""
    p, d = _stommel_problem(1.8e-11)
    S, fl = util.run_oracle(p, 5000, 1e-12, LEX)
    assert np.array_equal(S, d['S2']) and np.array_equal(fl, d['S2_flags'])
    assert fl[2] == 457 and '%e' % fl[1] == '3.339391e-13'


def test_stommel_s1_survey_pins(oracle):
    """beta = 0 case: loop count and scalars observed from the reference at survey time
    (BASELINE.md section 1)."""
    p, _ = _stommel_problem(0.0)
    S, fl = util.run_oracle(p, 5000, 1e-12, LEX)
    assert fl[2] == 3213 and '%e' % fl[1] == '9.916639e-13'
    assert '%.10e' % S.max() == '6.1120365308e+05'
    assert '%.10e' % np.abs(S).mean() == '2.7816492185e+05'


@pytest.mark.parametrize('tag,BCs', [('ep', ('extend', 'periodic')), ('fp', ('fixed', 'periodic'))])
def test_poisson_real_data(oracle, tag, BCs):
    """Data/Helmholtz_atmos.nc vorticity (reference tests/test_Poisson.py:14-24 inputs)."""
    from xinvert_amd import apps
    from xinvert_amd.field import Field
    d = golden('poisson_atmos.npz')
    lat, lon = d['lat'].astype(np.float64), d['lon'].astype(np.float64)
    Fv = Field(d['vor_f32'].astype(np.float64), ('time', 'lat', 'lon'), {'lat': lat, 'lon': lon})
    iP = apps._update(apps.default_iParams, {'BCs': list(BCs)})
    F, initS, (A, B, C) = apps._coeffs_Poisson(Fv, ['lat', 'lon'], 'lat-lon', apps.default_mParams, iP, None)
    ps = apps._cal_params2D(lat, lon, 'lat-lon')
    assert abs(ps['optArg'] - 1.934805014448645) < 1e-14
    for t in range(2):
        p = dict(kind='std2d', yc=73, xc=144, BCy=BCs[0], BCx=BCs[1], dely=ps['del2'], delx=ps['del1'],
                 delxSqr=ps['del1Sqr'], ratioQtr=ps['ratioQtr'], ratioSqr=ps['ratioSqr'],
                 optArg=ps['optArg'], undef=U, S0=np.zeros((73, 144)),
                 coefs=[A, B, C, np.ascontiguousarray(F.values[t])])
        S, fl = util.run_oracle(p, 59, 0.0, LEX)
        assert np.array_equal(S, d['S60_' + tag][t])
        assert np.array_equal(fl, d['flags_' + tag][t])


@pytest.mark.parametrize('tag,mx,tol,loops', [('hadley', 600, 1e-10, 142), ('tc', 600, 1e-12, 600)])
def test_eliassen_real_data_bitwise(oracle, tag, mx, tol, loops):
    p, d, ps = util.eliassen_problem(tag)
    if tag == 'hadley':
        assert p['optArg'] == ps['optArg']             # default over-relaxation factor
    S, fl = util.run_oracle(p, mx, tol, LEX)
    assert fl[2] == loops
    assert np.array_equal(S, d[tag + '_S']) and np.array_equal(fl, d[tag + '_flags'])


def test_norm_semantics(oracle):
    """absNorm2D (numbas.py:1710-1728): mean |S| over S != undef, NaN when nothing is defined."""
    S = np.array([[1.0, -3.0, U], [U, 2.0, -2.0]])
    assert oracle.abs_norm(S, U) == 2.0
    assert np.isnan(oracle.abs_norm(np.full((3, 4), U), U))


def test_sweeps_equal_flags2_plus_one(oracle):
    """Loop-count semantics: an un-converged run performs mxLoop + 1 sweeps."""
    p = util.rand2d('gen2d', 12, 16, 'fixed', 'fixed', seed=2)
    Sa, fa = util.run_oracle(p, 4, 0.0, LEX)                 # 5 sweeps
    q = dict(p)
    S1, f1 = util.run_oracle(p, 1, 0.0, LEX)                 # 2 sweeps
    q['S0'] = S1
    S2, f2 = util.run_oracle(q, 2, 0.0, LEX)                 # + 3 sweeps
    assert fa[2] == 4 and np.array_equal(Sa, S2)


# ------------------------------------------------------------------ biharmonic (Munk), SURVEY 8(f)
def test_bih_small_cases_bitwise(oracle):
    d = golden('bih_cases.npz')
    metas = [ast.literal_eval(str(m)) for m in d['meta']]
    assert len(metas) == 96
    for m in metas:
        k, yc, xc, BCy, BCx, dely, delx, om, nsw, tol = m
        arr = d[k + '_in']
        S = np.ascontiguousarray(arr[0]).copy()
        fl = np.array([0., 1., 0.])
        c = [np.ascontiguousarray(a) for a in arr[1:]]
        r = delx / dely
        oracle.general_bih_2d(S, *c, yc, xc, dely, delx, BCy, BCx, delx**4, delx**3, delx**2, r, r**4,
                              r / 4, r**2, om, U, fl, nsw, tol, LEX)
        assert np.array_equal(S, d[k + '_S']), m
        assert np.array_equal(fl, d[k + '_flags']), m


def munk_problem(A4):
    """reference tests/test_MunkWBC.py:14-58 inputs through the front end's coefficient code."""
    from xinvert_amd import apps
    from xinvert_amd.field import Field
    xnum, ynum = 201, 151
    Lx, Ly = 1e7, 2 * np.pi * 1e6
    x = np.linspace(0, Lx, xnum); y = np.linspace(0, Ly, ynum)
    curl = -0.3 * np.sin(np.pi * (y[:, None] + 0 * x[None, :]) / Ly) * np.pi / Ly
    iP = apps._update(apps.default_iParams, {'BCs': ['fixed', 'fixed'], 'optArg': 1.0})
    mP = apps._update(apps.default_mParams, {'A4': A4, 'beta': 1.8e-11, 'R': 0.0001, 'D': 200})
    F = Field(curl, ('ydef', 'xdef'), {'ydef': y, 'xdef': x})
    J, initS, cs = apps._coeffs_StommelMunk(F, ['ydef', 'xdef'], 'cartesian', mP, iP, None)
    ps = apps._cal_params2D(y, x, 'cartesian')
    return dict(kind='bih2d', yc=ynum, xc=xnum, BCy='fixed', BCx='fixed', dely=ps['del2'],
                delx=ps['del1'], delxSSr=ps['del1SSr'], delxTr=ps['del1Tr'], delxSqr=ps['del1Sqr'],
                ratio=ps['ratio'], ratioSSr=ps['ratioSSr'], ratioQtr=ps['ratioQtr'],
                ratioSqr=ps['ratioSqr'], optArg=1.0, undef=U, S0=np.zeros((ynum, xnum)),
                coefs=[np.ascontiguousarray(c) for c in cs] + [J.values]), curl, x, y


@pytest.mark.parametrize('A4,pin', [(5e3, 388730.8493746), (5e2, 399667.8611556)])
def test_munk_known_answers(oracle, A4, pin):
    """tests/test_MunkWBC.py:57-58 (np.isclose).  NB the first pin (A4 = 5e3) is an UN-converged
    iterate (loop 4000 of a Gauss-Seidel run still changing by 3e-5 per sweep): only the
    reference's own ordering can reproduce it."""
    p, _, _, _ = munk_problem(A4)
    S, fl = util.run_oracle(p, 4000, 1e-14, LEX)
    assert np.isclose(S.max(), pin) and abs(S.max() / pin - 1) < 1e-11
    assert (fl[2] == 4000) == (A4 == 5e3)


# ------------------------------------------------------------------ general 3-D (3DOcean), SURVEY 8(f)
def test_gen3d_small_cases_bitwise(oracle):
    d = golden('gen3d_cases.npz')
    metas = [ast.literal_eval(str(m)) for m in d['meta']]
    assert len(metas) == 32
    for m in metas:
        k, zc, yc, xc, BCy, BCx, delz, dely, delx, om, nsw, tol = m
        arr = d[k + '_in']
        S = np.ascontiguousarray(arr[0]).copy()
        fl = np.array([0., 1., 0.])
        c = [np.ascontiguousarray(a) for a in arr[1:]]
        oracle.general_3d(S, *c, zc, yc, xc, delz, dely, delx, 'fixed', BCy, BCx, delx**2,
                          delx / delz, delx / dely, (delx / delz)**2, (delx / dely)**2, om, U, fl,
                          nsw, tol, LEX)
        assert np.array_equal(S, d[k + '_S']), m
        assert np.array_equal(fl, d[k + '_flags']), m


# ------------------------------------------------------------------ standard_2D_test, SURVEY 8(f)
def test_std2dt_small_cases_bitwise(oracle):
    d = golden('std2dt_cases.npz')
    metas = [ast.literal_eval(str(m)) for m in d['meta']]
    assert len(metas) == 72
    for m in metas:
        k, yc, xc, BCy, BCx, dely, delx, om, nsw, tol = m
        arr = d[k + '_in']
        S = np.ascontiguousarray(arr[0]).copy()
        fl = np.array([0., 1., 0.])
        c = [np.ascontiguousarray(a) for a in arr[1:]]
        r = delx / dely
        oracle.standard_2d_test(S, *c, yc, xc, dely, delx, BCy, BCx, delx**2, r / 4, r**2, om, U, fl,
                                nsw, tol, LEX)
        assert np.array_equal(S, d[k + '_S']) and np.array_equal(fl, d[k + '_flags']), m


def fofonoff_problem():
    """reference tests/test_Fofonoff.py:13-41 / docs notebook 09 cell 2."""
    from xinvert_amd import apps
    from xinvert_amd.field import Field
    xc = np.linspace(0, 600000, 301); yc = np.linspace(0, 500000, 251)
    Fv = yc[:, None] - xc[None, :]
    iP = apps._update(apps.default_iParams, {'BCs': ['fixed', 'fixed'], 'optArg': 1.2})
    mP = apps._update(apps.default_mParams, {'f0': 1e-4, 'beta': 2e-11, 'c0': 8e-9, 'c1': 1e-4})
    F = Field(Fv, ('y', 'x'), {'y': yc, 'x': xc})
    Fm, initS, cs = apps._coeffs_Fofonoff(F, ['y', 'x'], 'cartesian', mP, iP, None)
    ps = apps._cal_params2D(yc, xc, 'cartesian')
    return dict(kind='std2dt', yc=251, xc=301, BCy='fixed', BCx='fixed', dely=ps['del2'], delx=ps['del1'],
                delxSqr=ps['del1Sqr'], ratio=ps['ratio'], ratioQtr=ps['ratioQtr'], ratioSqr=ps['ratioSqr'],
                optArg=1.2, undef=U, S0=np.zeros((251, 301)),
                coefs=[np.ascontiguousarray(c) for c in cs] + [Fm.values]), Fv, xc, yc


def test_fofonoff_notebook_pin(oracle):
    """docs/source/notebooks/09_Fofonoff_flow.ipynb:128 prints `loops 1174 and tolerance is
    9.362824e-15` for this call (mxLoop 4000, tolerance 1e-14, optArg 1.2)."""
    p, _, _, _ = fofonoff_problem()
    S, fl = util.run_oracle(p, 4000, 1e-14, LEX)
    assert '{0:4.0f} and tolerance is {1:e}'.format(fl[2], fl[1]) == '1174 and tolerance is 9.362824e-15'


def bretherton_problem():
    """reference tests/test_Bretherton.py:13-31: Data/topo.nc (201 x 301, here tests/golden/topo.npz --
    the file's three arrays), topography anomaly, f0 = 1e-4, D = 1000, lambda = 1e-15."""
    from xinvert_amd import apps
    from xinvert_amd.field import Field
    g = util.golden('topo.npz')
    topo = g['topo'] - g['topo'].mean()
    h = Field(topo, ('y', 'x'), {'y': g['y'], 'x': g['x']})
    iP = apps._update(apps.default_iParams, {'BCs': ['fixed', 'fixed'], 'mxLoop': 3000, 'tolerance': 1e-16})
    mP = apps._update(apps.default_mParams, {'f0': 1e-4, 'D': 1000, 'lambda': 1e-15})
    Fm, initS, cs = apps._coeffs_Bretherton(h, ['y', 'x'], 'cartesian', mP, iP, None)
    ps = apps._cal_params2D(g['y'], g['x'], 'cartesian')
    p = dict(kind='std2dt', yc=201, xc=301, BCy='fixed', BCx='fixed', dely=ps['del2'], delx=ps['del1'],
             delxSqr=ps['del1Sqr'], ratio=ps['ratio'], ratioQtr=ps['ratioQtr'], ratioSqr=ps['ratioSqr'],
             optArg=ps['optArg'], undef=U, S0=np.zeros((201, 301)),
             coefs=[np.ascontiguousarray(c) for c in cs] + [Fm.values])
    return p, h


def test_bretherton_ke_pin(oracle):
    """tests/test_Bretherton.py:42: `np.isclose(KE, 0.0812731)` with KE = sum(u^2 + v^2) / 2 of
    cal_flow(S1) after invert_BrethertonHaidvogel(topo) -- the reference's own known answer, reproduced
    by the oracle's lexicographic ordering through the host-side coefficient builder and cal_flow."""
    from xinvert_amd import apps
    p, h = bretherton_problem()
    S, fl = util.run_oracle(p, 3000, 1e-16, LEX)
    u, v = apps.cal_flow(h.like(S), ['y', 'x'], coords='cartesian')
    KE = float((u.values ** 2 + v.values ** 2).sum() / 2)
    assert np.isclose(KE, 0.0812731), KE
