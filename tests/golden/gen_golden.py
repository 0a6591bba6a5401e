#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE ITSELF.  Build container only.

Runs the reference's own kernels (/root/reference/xinvert/numbas.py, imported as plain Python
through oracle/ref_import.py) on seeded inputs and stores inputs + outputs as numeric
fixtures.  The committed fixtures are data only; this script and the reference tree never run
on the GPU box.  Re-run:  python tests/golden/gen_golden.py   (about 4 minutes).

Fixtures written
  small_cases.npz   randomized tiny grids: every (kernel x BCy x BCx x mask x B==0/B!=0) cell,
                    inputs, S after the lexicographic sweeps, flags
  bih_cases.npz     the same for the biharmonic kernel (numbas.invert_general_bih_2D)
  std2dt_cases.npz  the same for numbas.invert_standard_2D_test
  gen3d_cases.npz   the same for numbas.invert_general_3D
  gill_matsuno.npz  the reference's Gill-Matsuno known-answer case (tests/test_GillMatsuno.py:
                    14-57 inputs; notebook 07 parameters mxLoop=600, tol=1e-5): fields + flags
  stommel.npz       tests/test_StommelWBC.py:14-55 case S2 (beta = 1.8e-11): field + flags
  poisson_atmos.npz real data: Data/Helmholtz_atmos.nc `vor` (2x73x144 f32, promoted to f64),
                    lat, lon, and S after 60 reference sweeps for two BC sets
  eliassen.npz      real data for the 9-point form: Data/ZonalMean.nc (Hadley) and Data/TC2D.nc (TC)
                    coefficients + forcing, S and flags from the reference kernel
  mjo_ol.npz        real data: Data/MJO.nc `ol` (73x144 f32) for the Gill-Matsuno real case
"""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))

from oracle.ref_import import load_reference_numbas   # noqa: E402
from xinvert_amd import apps                            # noqa: E402
from xinvert_amd.field import Field                     # noqa: E402

U = -9.99e8
ref = load_reference_numbas()


def small_cases():
    rng = np.random.default_rng(20250509)
    out = {}
    meta = []
    cid = 0

    def mk(shape):
        return rng.uniform(0.5, 1.5, shape)

    grids2 = [(17, 24), (12, 19), (9, 30)]
    for (yc, xc) in grids2:
        for BCy in ('fixed', 'extend'):
            for BCx in ('fixed', 'periodic', 'extend'):
                for bnz in (0, 1):
                    for msk in (0, 1):
                        A = mk((yc, xc)); C = mk((yc, xc))
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
                        dely, delx = 1.3, 1.1
                        r = delx / dely
                        omega = 1.3
                        nsw = 25
                        # standard
                        S = S0.copy(); fl = np.array([0., 1., 0.])
                        ref.invert_standard_2D(S, A, B, C, F, yc, xc, dely, delx, BCy, BCx,
                                               delx**2, r / 4, r**2, omega, U, fl, nsw, 1e-9)
                        k = 'c%03d' % cid; cid += 1
                        out[k + '_in'] = np.stack([S0, A, B, C, F])
                        out[k + '_S'] = S; out[k + '_flags'] = fl
                        meta.append((k, 'std2d', yc, xc, BCy, BCx, dely, delx, omega, nsw, 1e-9))
                        # general
                        D = mk((yc, xc)) * 0.1; E = mk((yc, xc)) * 0.1
                        Fc = -mk((yc, xc)) * 0.01; G = F
                        S = S0.copy(); fl = np.array([0., 1., 0.])
                        ref.invert_general_2D(S, A, B, C, D, E, Fc, G, yc, xc, dely, delx, BCy, BCx,
                                              delx**2, r, r / 4, r**2, omega, U, fl, nsw, 1e-9)
                        k = 'c%03d' % cid; cid += 1
                        out[k + '_in'] = np.stack([S0, A, B, C, D, E, Fc, G])
                        out[k + '_S'] = S; out[k + '_flags'] = fl
                        meta.append((k, 'gen2d', yc, xc, BCy, BCx, dely, delx, omega, nsw, 1e-9))
    for (zc, yc, xc) in [(6, 9, 12), (5, 7, 9)]:
        for BCy in ('fixed', 'extend'):
            for BCx in ('fixed', 'periodic'):
                for msk in (0, 1):
                    sh = (zc, yc, xc)
                    A = mk(sh); B = mk(sh); C = mk(sh); F = rng.standard_normal(sh)
                    if msk:
                        F[rng.random(sh) < 0.15] = U
                        B[rng.random(sh) < 0.03] = U
                    S0 = rng.standard_normal(sh) * 0.1
                    if msk:
                        S0[rng.random(sh) < 0.05] = U
                    delz, dely, delx = 2.0, 1.3, 1.1
                    omega, nsw = 1.2, 15
                    S = S0.copy(); fl = np.array([0., 1., 0.])
                    ref.invert_standard_3D(S, A, B, C, F, zc, yc, xc, delz, dely, delx, 'fixed',
                                           BCy, BCx, delx**2, (delx / delz)**2, (delx / dely)**2,
                                           omega, U, fl, nsw, 1e-9)
                    k = 'c%03d' % cid; cid += 1
                    out[k + '_in'] = np.stack([S0, A, B, C, F])
                    out[k + '_S'] = S; out[k + '_flags'] = fl
                    meta.append((k, 'std3d', zc, yc, xc, BCy, BCx, delz, dely, delx, omega, nsw, 1e-9))
    out['meta'] = np.array([repr(m) for m in meta])
    np.savez_compressed(os.path.join(HERE, 'small_cases.npz'), **out)
    print('small_cases: %d cases' % cid)


def bih_cases():
    """numbas.invert_general_bih_2D (Munk / Stommel-Munk) on tiny random grids: every
    BCy x BCx x mask x (B,E == 0 | != 0) cell, including the periodic east branches."""
    rng = np.random.default_rng(20250510)
    out, meta, cid = {}, [], 0
    for (yc, xc) in [(9, 12), (8, 9), (11, 16), (7, 7)]:
        for BCy in ('fixed', 'extend'):
            for BCx in ('fixed', 'periodic', 'extend'):
                if yc > xc and BCy == 'extend' and BCx != 'periodic':
                    continue
                for bnz in (0, 1):
                    for msk in (0, 1):
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
                        dely, delx = 1.3, 1.1
                        r = delx / dely
                        S = S0.copy(); fl = np.array([0., 1., 0.])
                        ref.invert_general_bih_2D(S, A, B, C, D, E, F, G, H, I, J, yc, xc, dely, delx,
                                                  BCy, BCx, delx**4, delx**3, delx**2, r, r**4, r / 4,
                                                  r**2, 0.9, U, fl, 12, 1e-9)
                        k = 'b%03d' % cid; cid += 1
                        out[k + '_in'] = np.stack([S0, A, B, C, D, E, F, G, H, I, J])
                        out[k + '_S'] = S; out[k + '_flags'] = fl
                        meta.append((k, yc, xc, BCy, BCx, dely, delx, 0.9, 12, 1e-9))
    out['meta'] = np.array([repr(m) for m in meta])
    np.savez_compressed(os.path.join(HERE, 'bih_cases.npz'), **out)
    print('bih_cases: %d cases' % cid)


def std2dt_cases():
    """numbas.invert_standard_2D_test on tiny random grids."""
    rng = np.random.default_rng(20250511)
    out, meta, cid = {}, [], 0
    for (yc, xc) in [(9, 12), (8, 9), (11, 16)]:
        for BCy in ('fixed', 'extend'):
            for BCx in ('fixed', 'periodic', 'extend'):
                for bnz in (0, 1):
                    for msk in (0, 1):
                        sh = (yc, xc)
                        mk = lambda s=1.0: rng.uniform(0.5, 1.5, sh) * s
                        A, D = mk(), mk()
                        B = rng.uniform(-.2, .2, sh) if bnz else np.zeros(sh)
                        C = rng.uniform(-.2, .2, sh) if bnz else np.zeros(sh)
                        E = -mk(0.05); F = rng.standard_normal(sh)
                        if msk:
                            F[rng.random(sh) < 0.15] = U
                            A[rng.random(sh) < 0.03] = U
                        S0 = rng.standard_normal(sh) * 0.1
                        r = 1.1 / 1.3
                        S = S0.copy(); fl = np.array([0., 1., 0.])
                        ref.invert_standard_2D_test(S, A, B, C, D, E, F, yc, xc, 1.3, 1.1, BCy, BCx,
                                                    1.1**2, r / 4, r**2, 1.3, U, fl, 12, 1e-9)
                        k = 't%03d' % cid; cid += 1
                        out[k + '_in'] = np.stack([S0, A, B, C, D, E, F])
                        out[k + '_S'] = S; out[k + '_flags'] = fl
                        meta.append((k, yc, xc, BCy, BCx, 1.3, 1.1, 1.3, 12, 1e-9))
    out['meta'] = np.array([repr(m) for m in meta])
    np.savez_compressed(os.path.join(HERE, 'std2dt_cases.npz'), **out)
    print('std2dt_cases: %d cases' % cid)


def gill_matsuno():
    """Inputs as reference tests/test_GillMatsuno.py:14-40; iteration parameters as the executed
    notebook docs/source/notebooks/07_Gill_Matsuno_model.ipynb (mxLoop 600, tolerance 1e-5), whose
    printed `loops ... and tolerance is ...` lines are reproduced here by the reference code."""
    lonv = np.linspace(0, 360, 144); latv = np.linspace(-90, 90, 73)
    lat, lon = np.meshgrid(latv, lonv, indexing='ij')
    Q1 = 0.05 * np.exp(-((lat - 0)**2 + (lon - 120)**2) / 100.0)
    Q2 = 0.05 * np.exp(-((lat - 10)**2 + (lon - 120)**2) / 100.0) \
        - 0.05 * np.exp(-((lat + 10)**2 + (lon - 120)**2) / 100.0)
    Q3 = 0.05 * np.exp(-((lat - 10)**2 + (lon - 120)**2) / 100.0)
    out = {'lat': latv, 'lon': lonv}
    iP = apps._update(apps.default_iParams, {'BCs': ['fixed', 'periodic'], 'mxLoop': 600,
                                            'tolerance': 1e-5, 'optArg': 1.4})
    mP = apps._update(apps.default_mParams, {'epsilon': 1e-5, 'Phi': 5000})
    for name, Q in (('Q1', Q1), ('Q2', Q2), ('Q3', Q3)):
        Fq = Field(Q, ('lat', 'lon'), {'lat': latv, 'lon': lonv})
        G, initS, (A, B, C, D, E, Fc) = apps._coeffs_GillMatsuno(Fq, ['lat', 'lon'], 'lat-lon',
                                                                   mP, iP, None)
        ps = apps._cal_params2D(latv, lonv, 'lat-lon', Rearth=mP['Rearth'])
        S = np.zeros((73, 144)); fl = np.array([0., 1., 0.])
        t = time.time()
        ref.invert_general_2D(S, A, B, C, D, E, Fc, G.values, 73, 144, ps['del2'], ps['del1'],
                              'fixed', 'periodic', ps['del1Sqr'], ps['ratio'], ps['ratioQtr'],
                              ps['ratioSqr'], 1.4, U, fl, 600, 1e-5)
        print('GM %s: loops %4.0f and tolerance is %e  (%.1fs)' % (name, fl[2], fl[1], time.time() - t))
        out[name] = Q; out[name + '_S'] = S; out[name + '_flags'] = fl
    np.savez_compressed(os.path.join(HERE, 'gill_matsuno.npz'), **out)


def stommel():
    """reference tests/test_StommelWBC.py:14-55, case S2 (beta = 1.8e-11)."""
    xnum, ynum = 201, 151
    Lx, Ly = 1e7, 2 * np.pi * 1e6
    R, depth, beta, Fw = 0.0008, 200, 1.8e-11, 0.3
    xdef = np.linspace(0, Lx, xnum); ydef = np.linspace(0, Ly, ynum)
    ygrid, xgrid = np.meshgrid(ydef, xdef, indexing='ij')
    curl = -Fw * np.sin(np.pi * ygrid / Ly) * np.pi / Ly
    iP = apps._update(apps.default_iParams, {'BCs': ['fixed', 'fixed'], 'mxLoop': 5000,
                                            'optArg': 1.9, 'tolerance': 1e-12})
    mP = apps._update(apps.default_mParams, {'beta': beta, 'R': R, 'D': depth})
    Fc_ = Field(curl, ('ydef', 'xdef'), {'ydef': ydef, 'xdef': xdef})
    G, initS, (A, B, C, D, E, Fc) = apps._coeffs_Stommel(Fc_, ['ydef', 'xdef'], 'cartesian', mP, iP, None)
    ps = apps._cal_params2D(ydef, xdef, 'cartesian')
    S = np.zeros((ynum, xnum)); fl = np.array([0., 1., 0.])
    t = time.time()
    ref.invert_general_2D(S, A, B, C, D, E, Fc, G.values, ynum, xnum, ps['del2'], ps['del1'],
                          'fixed', 'fixed', ps['del1Sqr'], ps['ratio'], ps['ratioQtr'],
                          ps['ratioSqr'], 1.9, U, fl, 5000, 1e-12)
    print('Stommel S2: loops %4.0f and tolerance is %e max %.10e meanabs %.10e (%.1fs)'
          % (fl[2], fl[1], S.max(), np.abs(S).mean(), time.time() - t))
    np.savez_compressed(os.path.join(HERE, 'stommel.npz'), ydef=ydef, xdef=xdef, curl=curl,
                        S2=S, S2_flags=fl)


def real_data():
    """Extract the bundled sample fields (HDF5) with the conda python that has h5py, then run
    the reference Poisson kernel on the float64-promoted vorticity."""
    tmp = os.path.join(HERE, '_tmp_extract.npz')
    code = (
        "import h5py, numpy as np\n"
        "f=h5py.File('/root/reference/Data/Helmholtz_atmos.nc','r')\n"
        "g=h5py.File('/root/reference/Data/MJO.nc','r')\n"
        "np.savez(%r, vor=f['vor'][...], lat=f['lat'][...], lon=f['lon'][...],"
        " ol=g['ol'][...], mlat=g['lat'][...], mlon=g['lon'][...])\n" % tmp)
    subprocess.check_call(['/opt/conda/bin/python3.9', '-c', code])
    d = np.load(tmp)
    os.remove(tmp)
    vor = d['vor'].astype(np.float64); lat = d['lat'].astype(np.float64); lon = d['lon'].astype(np.float64)
    out = {'vor_f32': d['vor'], 'lat': d['lat'], 'lon': d['lon']}
    Fv = Field(vor, ('time', 'lat', 'lon'), {'lat': lat, 'lon': lon})
    for tag, BCs in (('ep', ['extend', 'periodic']), ('fp', ['fixed', 'periodic'])):
        iP = apps._update(apps.default_iParams, {'BCs': BCs})
        F, initS, (A, B, C) = apps._coeffs_Poisson(Fv, ['lat', 'lon'], 'lat-lon',
                                                   apps.default_mParams, iP, None)
        ps = apps._cal_params2D(lat, lon, 'lat-lon')
        res = []
        fls = []
        for t in range(2):
            S = np.zeros((73, 144)); fl = np.array([0., 1., 0.])
            ref.invert_standard_2D(S, A, B, C, np.ascontiguousarray(F.values[t]), 73, 144,
                                   ps['del2'], ps['del1'], BCs[0], BCs[1], ps['del1Sqr'],
                                   ps['ratioQtr'], ps['ratioSqr'], ps['optArg'], U, fl, 59, 0.0)
            res.append(S); fls.append(fl)
        out['S60_' + tag] = np.stack(res); out['flags_' + tag] = np.stack(fls)
        print('poisson_atmos', tag, fls[0], fls[1])
    np.savez_compressed(os.path.join(HERE, 'poisson_atmos.npz'), **out)
    np.savez_compressed(os.path.join(HERE, 'mjo_ol.npz'), ol_f32=d['ol'], lat=d['mlat'], lon=d['mlon'])
def eliassen():
    """Real data for the 9-point standard form (B != 0): the reference's Eliassen tests.
    Hadley: tests/test_Eliassen.py:16-147 inverts F_EHF + F_AHF with Acoef/Bcoef/Ccoef -- exactly
    the fields bundled in Data/ZonalMean.nc (37 x 72, f64), BCs fixed/fixed, mxLoop 600, tol 1e-10.
    TC: tests/test_Eliassen.py:207-232 (Data/TC2D.nc Aa/Bb/Cc/faf, 37 x 50 f32 promoted to f64,
    undef 9.99e20 -> NaN, optArg 1.4, tol 1e-12)."""
    tmp = os.path.join(HERE, '_tmp_extract.npz')
    code = (
        "import h5py, numpy as np\n"
        "z=h5py.File('/root/reference/Data/ZonalMean.nc','r')\n"
        "t=h5py.File('/root/reference/Data/TC2D.nc','r')\n"
        "np.savez(%r, zA=z['Acoef'][...], zB=z['Bcoef'][...], zC=z['Ccoef'][...], zEHF=z['EHF'][...],"
        " zEAF=z['EAF'][...], zlev=z['LEV'][...], zlat=z['lat'][...], tA=t['Aa'][...], tB=t['Bb'][...],"
        " tC=t['Cc'][...], tF=t['faf'][...], tlev=t['lev'][...], tlat=t['lat'][...])\n" % tmp)
    subprocess.check_call(['/opt/conda/bin/python3.9', '-c', code])
    d = np.load(tmp)
    os.remove(tmp)
    out = {}
    und = np.float32(9.99e20)
    cases = {
        'hadley': dict(A=d['zA'], B=d['zB'], C=d['zC'], F=d['zEHF'] + d['zEAF'], lev=d['zlev'],
                       lat=d['zlat'].astype(np.float64), iP={'mxLoop': 600, 'tolerance': 1e-10}),
        'tc': dict(A=np.where(d['tA'] != und, d['tA'], np.nan).astype(np.float64),
                   B=np.where(d['tB'] != und, d['tB'], np.nan).astype(np.float64),
                   C=np.where(d['tC'] != und, d['tC'], np.nan).astype(np.float64),
                   F=np.where(d['tF'] != und, d['tF'], np.nan).astype(np.float64),
                   lev=d['tlev'], lat=d['tlat'], iP={'mxLoop': 600, 'tolerance': 1e-12, 'optArg': 1.4}),
    }
    for tag, c in cases.items():
        lev, lat = c['lev'], c['lat']
        F = Field(c['F'], ('lev', 'lat'), {'lev': lev, 'lat': lat})
        iP = apps._update(apps.default_iParams, dict(c['iP'], BCs=['fixed', 'fixed']))
        mP = apps._update(apps.default_mParams, {'A': c['A'], 'B': c['B'], 'C': c['C']},
                          ['A', 'B', 'C', 'g', 'Omega', 'Rearth'])
        Fm, initS, (A, B, C) = apps._coeffs_Eliassen(F, ['lev', 'lat'], 'z-lat', mP, iP, None)
        ps = apps._cal_params2D(lev, lat, 'z-lat')
        om = iP['optArg'] if iP['optArg'] is not None else ps['optArg']
        yc, xc = F.shape
        S = np.zeros((yc, xc)); fl = np.array([0., 1., 0.])
        t = time.time()
        ref.invert_standard_2D(S, np.ascontiguousarray(A), np.ascontiguousarray(B), np.ascontiguousarray(C),
                               np.ascontiguousarray(Fm.values), yc, xc, ps['del2'], ps['del1'], 'fixed',
                               'fixed', ps['del1Sqr'], ps['ratioQtr'], ps['ratioSqr'], om, U, fl,
                               iP['mxLoop'], iP['tolerance'])
        print('Eliassen %s: loops %4.0f and tolerance is %e, max|S| %.6e, nan %d (%.1fs)'
              % (tag, fl[2], fl[1], np.nanmax(np.abs(S)), np.isnan(S).sum(), time.time() - t))
        for k in ('A', 'B', 'C', 'F', 'lev', 'lat'):
            out[tag + '_' + k] = c[k]
        out[tag + '_S'] = S; out[tag + '_flags'] = fl
        out[tag + '_optArg'] = np.float64(om)
    np.savez_compressed(os.path.join(HERE, 'eliassen.npz'), **out)


def gen3d_cases():
    """numbas.invert_general_3D (3DOcean) on tiny random volumes: BCy x BCx x mask, including
    yc > xc with periodic x and a masked H at i == 0 (the west branch never tests H)."""
    rng = np.random.default_rng(20250512)
    out, meta, cid = {}, [], 0
    for (zc, yc, xc) in [(6, 9, 12), (5, 7, 9), (4, 10, 8)]:
        for BCy in ('fixed', 'extend'):
            for BCx in ('fixed', 'periodic', 'extend'):
                if yc > xc and BCy == 'extend' and BCx != 'periodic':
                    continue                       # the reference indexes out of bounds there
                for msk in (0, 1):
                    sh = (zc, yc, xc)
                    mk = lambda s=1.0: rng.uniform(0.5, 1.5, sh) * s
                    A, B, C = mk(), mk(), mk()
                    D, E, F = mk(0.1), -mk(0.1), mk(0.1)
                    G = -mk(0.01); H = rng.standard_normal(sh)
                    if msk:
                        H[rng.random(sh) < 0.15] = U
                        H[:, :, 0][rng.random((zc, yc)) < 0.3] = U
                        B[rng.random(sh) < 0.03] = U
                        G[rng.random(sh) < 0.03] = U
                    S0 = rng.standard_normal(sh) * 0.1
                    if msk:
                        S0[rng.random(sh) < 0.05] = U
                    delz, dely, delx = 2.0, 1.3, 1.1
                    omega, nsw = 1.2, 12
                    S = S0.copy(); fl = np.array([0., 1., 0.])
                    ref.invert_general_3D(S, A, B, C, D, E, F, G, H, zc, yc, xc, delz, dely, delx,
                                          'fixed', BCy, BCx, delx**2, delx / delz, delx / dely,
                                          (delx / delz)**2, (delx / dely)**2, omega, U, fl, nsw, 1e-9)
                    k = 'g%03d' % cid; cid += 1
                    out[k + '_in'] = np.stack([S0, A, B, C, D, E, F, G, H])
                    out[k + '_S'] = S; out[k + '_flags'] = fl
                    meta.append((k, zc, yc, xc, BCy, BCx, delz, dely, delx, omega, nsw, 1e-9))
    out['meta'] = np.array([repr(m) for m in meta])
    np.savez_compressed(os.path.join(HERE, 'gen3d_cases.npz'), **out)
    print('gen3d_cases: %d cases' % cid)


if __name__ == '__main__':
    which = sys.argv[1:] or ['small', 'bih', 'std2dt', 'gen3d', 'real', 'eliassen', 'gm', 'stommel']
    if 'bih' in which: bih_cases()
    if 'std2dt' in which: std2dt_cases()
    if 'gen3d' in which: gen3d_cases()
    if 'small' in which: small_cases()
    if 'real' in which: real_data()
    if 'eliassen' in which: eliassen()
    if 'gm' in which: gill_matsuno()
    if 'stommel' in which: stommel()
