"""The coloured ordering (what the HIP kernels run) against the reference's lexicographic one.

Iterates differ sweep by sweep; the fixed point is the same.  Bar (north_star): converged
field within 1e-6 rel-L2 of the reference ordering.
"""
import numpy as np
import pytest

import util
from util import golden, U

LEX, AUTO, C2, C4 = 0, 1, 2, 4


def test_converged_gill_matsuno(oracle):
    from test_oracle_golden import _gm_problem
    d = golden('gill_matsuno.npz')
    for name in ('Q1', 'Q2', 'Q3'):
        p, _ = _gm_problem(d[name], d['lat'], d['lon'], 4000, 1e-13)
        Sl, fl = util.run_oracle(p, 4000, 1e-13, LEX)
        Sc, fc = util.run_oracle(p, 4000, 1e-13, AUTO)
        assert fl[2] < 4000 and fc[2] < 4000
        assert util.rel_l2(Sc, Sl) < 1e-6


def test_converged_stommel(oracle):
    from test_oracle_golden import _stommel_problem
    for beta in (0.0, 1.8e-11):
        p, _ = _stommel_problem(beta)
        Sl, fl = util.run_oracle(p, 8000, 1e-14, LEX)
        Sc, fc = util.run_oracle(p, 8000, 1e-14, AUTO)
        assert util.rel_l2(Sc, Sl) < 1e-6


def test_converged_ishida_mask_odd_periodic(oracle):
    """reference tests/test_Ishida.py:13-63: land strips (undef), periodic x with odd xc = 251
    -> the seam colours.  Also checks the reference's own bounds (:61-62)."""
    from xinvert_amd import apps
    from xinvert_amd.field import Field
    xnum, ynum = 251, 151
    Lx, Ly = 1e7, 2 * np.pi * 1e6
    x = np.linspace(0, Lx, xnum); y = np.linspace(0, Ly, ynum)
    yg = y[:, None] + 0 * x[None, :]
    curl = -np.pi * np.sin(2. * np.pi * yg / Ly) / Ly
    curl[65:, 100:104] = -9999
    curl[:75, 130:134] = -9999
    for R, bound in ((0.0009, 5.5e5), (0.0009 * 20, 2.8e4)):
        iP = apps._update(apps.default_iParams, {'BCs': ['fixed', 'periodic'], 'undef': -9999})
        mP = apps._update(apps.default_mParams, {'beta': 2.2e-11, 'R': R, 'D': 200})
        F = Field(curl, ('ydef', 'xdef'), {'ydef': y, 'xdef': x})
        G, initS, cs = apps._coeffs_Stommel(F, ['ydef', 'xdef'], 'cartesian', mP, iP, None)
        ps = apps._cal_params2D(y, x, 'cartesian')
        p = dict(kind='gen2d', yc=ynum, xc=xnum, BCy='fixed', BCx='periodic', dely=ps['del2'],
                 delx=ps['del1'], delxSqr=ps['del1Sqr'], ratio=ps['ratio'], ratioQtr=ps['ratioQtr'],
                 ratioSqr=ps['ratioSqr'], optArg=1.4, undef=U, S0=np.zeros((ynum, xnum)),
                 coefs=list(cs) + [G.values])
        Sl, fl = util.run_oracle(p, 3000, 1e-9, LEX)
        assert (np.abs(Sl) <= bound).all()
        Sl, _ = util.run_oracle(p, 20000, 1e-14, LEX)
        Sc, _ = util.run_oracle(p, 20000, 1e-14, AUTO)
        assert util.rel_l2(Sc, Sl) < 1e-6
        assert (Sc[G.values == U] == 0).all()          # land never updated


def test_converged_poisson_fixed_periodic(oracle):
    from xinvert_amd import synthetic
    p = synthetic.member(synthetic.poisson_latlon(90, 180, mask=True), 0)
    Sl, fl = util.run_oracle(p, 6000, 1e-14, LEX)
    Sc, fc = util.run_oracle(p, 6000, 1e-14, AUTO)
    assert util.rel_l2(Sc, Sl) < 1e-6


def test_converged_nine_point_four_colour(oracle):
    """B != 0 (true 9-point coupling): the 4-colour ordering converges to the same field."""
    p = util.rand2d('std2d', 30, 41, 'fixed', 'periodic', bnz=True, msk=True, seed=3, omega=1.2)
    Sl, _ = util.run_oracle(p, 5000, 1e-15, LEX)
    Sc, _ = util.run_oracle(p, 5000, 1e-15, AUTO)
    m = Sl != U
    assert util.rel_l2(Sc, Sl, m) < 1e-6
    q = util.rand2d('gen2d', 30, 40, 'extend', 'fixed', bnz=True, msk=False, seed=4, omega=1.2)
    Sl, _ = util.run_oracle(q, 5000, 1e-15, LEX)
    Sc, _ = util.run_oracle(q, 5000, 1e-15, AUTO)
    assert util.rel_l2(Sc, Sl) < 1e-6


def test_converged_3d(oracle):
    from xinvert_amd import synthetic
    p = synthetic.member(synthetic.omega_latlon(8, 24, 36, 1), 0)
    Sl, _ = util.run_oracle(p, 4000, 1e-15, LEX)
    Sc, _ = util.run_oracle(p, 4000, 1e-15, AUTO)
    assert util.rel_l2(Sc, Sl) < 1e-6


@pytest.mark.parametrize('BCx', ['fixed', 'periodic'])
def test_converged_general_3d(oracle, BCx):
    """numbas.invert_general_3D: red-black and lexicographic orders converge to the same field
    (odd xc with periodic x exercises the seam colours)."""
    p = util.rand3dg(6, 11, 15, 'fixed', BCx, msk=True, seed=31)
    p['S0'] = np.zeros_like(p['S0'])
    Sl, fl = util.run_oracle(p, 4000, 1e-15, LEX)
    Sc, fc = util.run_oracle(p, 4000, 1e-15, AUTO)
    assert fl[2] < 4000 and fc[2] < 4000
    assert util.rel_l2(Sc, Sl) < 1e-6


def test_auto_picks_two_colours_when_B_is_zero(oracle):
    p = util.rand2d('gen2d', 14, 20, 'fixed', 'periodic', bnz=False, msk=True, seed=8)
    Sa, fa = util.run_oracle(p, 10, 0.0, AUTO)
    S2, f2 = util.run_oracle(p, 10, 0.0, C2)
    S4, f4 = util.run_oracle(p, 10, 0.0, C4)
    assert np.array_equal(Sa, S2) and not np.array_equal(S2, S4)


def test_linearity_of_one_sweep(oracle):
    """One sweep is affine in (S, forcing): sweep(a S1 + b S2; a G1 + b G2) = a sweep(S1;G1) + b sweep(S2;G2)."""
    p1 = util.rand2d('gen2d', 16, 22, 'fixed', 'periodic', seed=1)
    p2 = dict(p1); rng = np.random.default_rng(9)
    p2['S0'] = rng.standard_normal(p1['S0'].shape)
    p2['coefs'] = list(p1['coefs']); p2['coefs'][-1] = rng.standard_normal(p1['S0'].shape)
    p3 = dict(p1)
    p3['S0'] = 2.0 * p1['S0'] - 0.5 * p2['S0']
    p3['coefs'] = list(p1['coefs']); p3['coefs'][-1] = 2.0 * p1['coefs'][-1] - 0.5 * p2['coefs'][-1]
    S1, _ = util.run_oracle(p1, 0, 0.0, C2)
    S2, _ = util.run_oracle(p2, 0, 0.0, C2)
    S3, _ = util.run_oracle(p3, 0, 0.0, C2)
    assert np.allclose(S3, 2.0 * S1 - 0.5 * S2, rtol=1e-11, atol=1e-11)


def test_converged_munk_nine_colours(oracle):
    """Biharmonic form, 9-colour ordering: the converged case of tests/test_MunkWBC.py."""
    from test_oracle_golden import munk_problem
    p, _, _, _ = munk_problem(5e2)
    Sl, fl = util.run_oracle(p, 4000, 1e-14, LEX)
    Sc, fc = util.run_oracle(p, 4000, 1e-14, AUTO)
    assert fl[2] < 4000 and fc[2] < 4000
    assert util.rel_l2(Sc, Sl) < 1e-6
    assert np.isclose(Sc.max(), 399667.8611556)
    q = util.randbih(14, 17, 'fixed', 'periodic', bnz=True, msk=True, seed=5)   # xc % 3 = 2: trailing colours
    Sl, _ = util.run_oracle(q, 20000, 1e-15, LEX)
    Sc, _ = util.run_oracle(q, 20000, 1e-15, AUTO)
    assert util.rel_l2(Sc, Sl, Sl != U) < 1e-6


def test_converged_fofonoff(oracle):
    from test_oracle_golden import fofonoff_problem
    p, _, _, _ = fofonoff_problem()
    Sl, fl = util.run_oracle(p, 4000, 1e-14, LEX)
    Sc, fc = util.run_oracle(p, 4000, 1e-14, AUTO)
    assert fc[2] < 4000 and util.rel_l2(Sc, Sl) < 1e-6
