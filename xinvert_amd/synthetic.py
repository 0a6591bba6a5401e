"""Seeded synthetic inputs for the benchmark configurations.

numpy only; the arrays go through the same `apps._coeffs_*` / `_cal_params*` code as user
data, so the kernels see exactly what the front end would hand them.
"""
import numpy as np

from . import apps
from .field import Field

SEED = 20250509


def _modes(rng, lat, lon, nmodes, kmax, lmax):
    """Smooth random field: sum of separable low-wavenumber modes, periodic in lon."""
    y = (lat - lat[0]) / (lat[-1] - lat[0])
    x = np.deg2rad(lon) if lon[-1] > 7 else 2 * np.pi * (lon - lon[0]) / (lon[-1] - lon[0] + (lon[1] - lon[0]))
    Y = np.empty((nmodes, lat.size)); X = np.empty((nmodes, lon.size))
    for m in range(nmodes):
        k = rng.integers(1, kmax + 1); l = rng.integers(1, lmax + 1)
        a = rng.standard_normal() / np.sqrt(nmodes)
        ph, ps = rng.uniform(0, 2 * np.pi, 2)
        Y[m] = a * np.sin(np.pi * l * y + ps)
        X[m] = np.cos(k * x + ph)
    # (an explicit loop over the modes, not `Y.T @ X`: a BLAS matmul adds in an order that depends on its thread count --
    #  torchrun sets OMP_NUM_THREADS=1 for its ranks -- and a member must hold the same bits in every process of a job:
    #  bench.py's S_checksum_sha256 compares the fields of differently split batches)
    out = np.zeros((lat.size, lon.size))
    for m in range(nmodes):
        out += Y[m][:, None] * X[m][None, :]
    return out


def poisson_latlon(ny, nx, mask=True, seed=SEED, BCs=('fixed', 'periodic'), members=1):
    """Config 1 / 2: invert_Poisson on a global lat-lon grid (ny x nx), optional land mask
    (mask=True: continent-size blobs; mask='coastline': the same plus coastline-scale structure).

    Returns dict(kind='std2d', S0 [m,ny,nx], coefs=[A,B,C (shared), F [m,ny,nx]], scalars...)."""
    rng = np.random.default_rng(seed)
    dlat, dlon = 180.0 / ny, 360.0 / nx
    lat = -90.0 + dlat / 2 + dlat * np.arange(ny)
    lon = dlon * np.arange(nx)
    zs = []
    for m in range(members):
        z = 1e-5 * _modes(rng, lat, lon, 16, 12, 10)
        w = np.cos(np.deg2rad(lat))[:, None]
        z -= (z * w).sum() / (w.sum() * nx)                   # zero area mean
        zs.append(z)
    zeta = np.stack(zs)
    if mask:
        land = _modes(rng, lat, lon, 16, 6, 5)
        if mask == 'coastline':
            # coastline-scale structure on top of the continents (wavenumbers up to 60) and a few hundred small
            # islands: few wave-tiles are land throughout, so masked-tile skipping finds little to skip.
            # (Drawn from a second generator: the default mask above stays what the committed fixtures saw.)
            rng2 = np.random.default_rng(seed + 1000)
            land = land + 0.5 * _modes(rng2, lat, lon, 64, 60, 50)
        thr = np.quantile(land[::7, ::7], 0.70)
        land_mask = (land > thr) | (np.abs(lat)[:, None] > 85.0)
        if mask == 'coastline':
            jj, ii = np.arange(ny)[:, None], np.arange(nx)[None, :]
            for _ in range(300):
                j0, i0, r = rng2.integers(0, ny), rng2.integers(0, nx), rng2.integers(2, 7) * max(1, ny // 900)
                di = np.minimum(np.abs(ii - i0), nx - np.abs(ii - i0))
                land_mask |= ((jj - j0) ** 2 + di ** 2) <= r * r
        zeta[:, land_mask] = np.nan
    F = Field(zeta, ('member', 'lat', 'lon'), {'lat': lat, 'lon': lon})
    iP = apps._update(apps.default_iParams, {'BCs': list(BCs)})
    Fm, initS, (A, B, C) = apps._coeffs_Poisson(F, ['lat', 'lon'], 'lat-lon', apps.default_mParams, iP, None)
    ps = apps._cal_params2D(lat, lon, 'lat-lon')
    return dict(kind='std2d', yc=ny, xc=nx, BCy=BCs[0], BCx=BCs[1], dely=ps['del2'], delx=ps['del1'],
                delxSqr=ps['del1Sqr'], ratio=ps['ratio'], ratioQtr=ps['ratioQtr'],
                ratioSqr=ps['ratioSqr'], optArg=ps['optArg'], undef=apps._undeftmp,
                S0=initS.values, coefs=[A, B, C, Fm.values], shared=(0, 1, 2), lat=lat, lon=lon,
                zeta=zeta)          # (the raw forcing, NaN on land: what a caller of apps.invert_Poisson holds)


def stommel_cartesian(ny, nx, seed=SEED, varying_R=True):
    """Config 3: Stommel gyre on a Cartesian box with spatially varying friction R(x, y)."""
    rng = np.random.default_rng(seed)
    Lx = Ly = 1e7
    x = np.linspace(0, Lx, nx); y = np.linspace(0, Ly, ny)
    yg = y[:, None] + 0 * x[None, :]
    curl = -0.3 * np.sin(np.pi * yg / Ly) * np.pi / Ly * (1.0 + 0.1 * rng.standard_normal((ny, nx)))
    if varying_R:
        s = _modes(rng, y, x, 8, 3, 3)
        R = 8e-4 * (1.0 + 0.5 * s / np.abs(s).max())
    else:
        R = 8e-4
    F = Field(curl, ('ydef', 'xdef'), {'ydef': y, 'xdef': x})
    iP = apps._update(apps.default_iParams, {'BCs': ['fixed', 'fixed']})
    mP = dict(apps.default_mParams); mP.update({'beta': 1.8e-11, 'R': R, 'D': 200, 'rho0': 1027})
    G, initS, cs = apps._coeffs_Stommel(F, ['ydef', 'xdef'], 'cartesian', mP, iP, None)
    ps = apps._cal_params2D(y, x, 'cartesian')
    return dict(kind='gen2d', yc=ny, xc=nx, BCy='fixed', BCx='fixed', dely=ps['del2'], delx=ps['del1'],
                delxSqr=ps['del1Sqr'], ratio=ps['ratio'], ratioQtr=ps['ratioQtr'],
                ratioSqr=ps['ratioSqr'], optArg=ps['optArg'], undef=apps._undeftmp,
                S0=initS.values[None], coefs=list(cs) + [G.values[None]], shared=(0, 1, 2, 3, 4, 5))


def munk_cartesian(ny, nx, seed=SEED, varying=False):
    """Config 3 (Munk branch): Stommel-Munk gyre, biharmonic form, Cartesian box.  varying=True: the lateral viscosity
    A4(x, y) and the bottom friction R(x, y) are smooth 2-D fields (BASELINE configs[2]: "spatially-varying" coefficients:
    A, C, D, F of the biharmonic form then vary along x)."""
    rng = np.random.default_rng(seed)
    Lx, Ly = 1e7, 2 * np.pi * 1e6
    x = np.linspace(0, Lx, nx); y = np.linspace(0, Ly, ny)
    yg = y[:, None] + 0 * x[None, :]
    curl = -0.3 * np.sin(np.pi * yg / Ly) * np.pi / Ly * (1.0 + 0.1 * rng.standard_normal((ny, nx)))
    F = Field(curl, ('ydef', 'xdef'), {'ydef': y, 'xdef': x})
    iP = apps._update(apps.default_iParams, {'BCs': ['fixed', 'fixed']})
    A4, R = 5e2 * (151.0 / ny) ** 0, 1e-4
    if varying:                                              # (drawn behind the forcing's noise: the constant case keeps its numbers)
        sa, sr = _modes(rng, y, x, 8, 3, 3), _modes(rng, y, x, 8, 3, 3)
        A4 = A4 * (1.0 + 0.4 * sa / np.abs(sa).max())
        R = R * (1.0 + 0.5 * sr / np.abs(sr).max())
    mP = dict(apps.default_mParams); mP.update({'A4': A4, 'beta': 1.8e-11, 'R': R, 'D': 200})
    J, initS, cs = apps._coeffs_StommelMunk(F, ['ydef', 'xdef'], 'cartesian', mP, iP, None)
    ps = apps._cal_params2D(y, x, 'cartesian')
    return dict(kind='bih2d', yc=ny, xc=nx, BCy='fixed', BCx='fixed', dely=ps['del2'], delx=ps['del1'],
                delxSSr=ps['del1SSr'], delxTr=ps['del1Tr'], delxSqr=ps['del1Sqr'], ratio=ps['ratio'],
                ratioSSr=ps['ratioSSr'], ratioQtr=ps['ratioQtr'], ratioSqr=ps['ratioSqr'],
                optArg=1.0, undef=apps._undeftmp, S0=initS.values[None],
                coefs=[np.ascontiguousarray(c) for c in cs] + [J.values[None]],
                shared=tuple(range(9)))


def gill_matsuno(ny, nx, members, seed=SEED):
    """Config 4: Gill-Matsuno response to `members` Gaussian heat sources."""
    rng = np.random.default_rng(seed)
    lat = np.linspace(-90, 90, ny); lon = np.linspace(0, 360, nx)
    la, lo = np.meshgrid(lat, lon, indexing='ij')
    Q = np.stack([0.05 * np.exp(-((la - rng.uniform(-20, 20))**2 + (lo - rng.uniform(60, 300))**2) / 100.0)
                  for _ in range(members)])
    F = Field(Q, ('member', 'lat', 'lon'), {'lat': lat, 'lon': lon})
    iP = apps._update(apps.default_iParams, {'BCs': ['fixed', 'periodic']})
    mP = dict(apps.default_mParams); mP.update({'epsilon': 1e-5, 'Phi': 5000})
    G, initS, cs = apps._coeffs_GillMatsuno(F, ['lat', 'lon'], 'lat-lon', mP, iP, None)
    ps = apps._cal_params2D(lat, lon, 'lat-lon')
    return dict(kind='gen2d', yc=ny, xc=nx, BCy='fixed', BCx='periodic', dely=ps['del2'], delx=ps['del1'],
                delxSqr=ps['del1Sqr'], ratio=ps['ratio'], ratioQtr=ps['ratioQtr'],
                ratioSqr=ps['ratioSqr'], optArg=1.4, undef=apps._undeftmp,
                S0=initS.values, coefs=list(cs) + [G.values], shared=(0, 1, 2, 3, 4, 5),
                lat=lat, lon=lon)


def omega_latlon(nz, ny, nx, steps=1, seed=SEED):
    """Config 5: QG omega equation with a topography mask."""
    rng = np.random.default_rng(seed)
    lev = np.linspace(1e5, 1e4, nz)
    dlat = 180.0 / ny
    lat = -90 + dlat / 2 + dlat * np.arange(ny); lon = (360.0 / nx) * np.arange(nx)
    N2 = 1e-6 * (1.0 + 4.0 * np.linspace(0, 1, nz)**2)
    frc = []
    for _ in range(steps):
        base = _modes(rng, lat, lon, 8, 6, 5)
        prof = np.sin(np.pi * np.linspace(0, 1, nz))[:, None, None]
        frc.append(1e-17 * prof * base[None])
    frc = np.stack(frc)
    ps_field = 1e5 - 3e4 * np.clip(_modes(rng, lat, lon, 8, 4, 4), 0, None)
    below = lev[:, None, None] > ps_field[None]
    frc[:, below] = np.nan
    F = Field(frc, ('time', 'lev', 'lat', 'lon'), {'lev': lev, 'lat': lat, 'lon': lon})
    iP = apps._update(apps.default_iParams, {'BCs': ['fixed', 'fixed', 'periodic']})
    mP = dict(apps.default_mParams); mP['N2'] = N2
    Fm, initS, (A, B, C) = apps._coeffs_omega(F, ['lev', 'lat', 'lon'], 'lat-lon', mP, iP, None)
    p3 = apps._cal_params3D(lev, lat, lon, 'lat-lon')
    return dict(kind='std3d', zc=nz, yc=ny, xc=nx, BCz='fixed', BCy='fixed', BCx='periodic',
                delz=p3['del3'], dely=p3['del2'], delx=p3['del1'], delxSqr=p3['del1Sqr'],
                ratio2Sqr=p3['ratio2Sqr'], ratio1Sqr=p3['ratio1Sqr'], optArg=p3['optArg'],
                undef=apps._undeftmp, S0=initS.values, coefs=[A, B, C, Fm.values], shared=(0, 1, 2))


def ocean3d_latlon(nz, ny, nx, steps=1, seed=SEED):
    """3-D wind-driven ocean (general 3-D form, apps.invert_3DOcean) with a bathymetry mask."""
    rng = np.random.default_rng(seed)
    lev = np.linspace(0.0, 500.0, nz)
    dlat = 120.0 / ny
    lat = -60 + dlat / 2 + dlat * np.arange(ny); lon = (360.0 / nx) * np.arange(nx)
    frc = []
    for _ in range(steps):
        base = _modes(rng, lat, lon, 8, 6, 5)
        prof = np.exp(-np.linspace(0, 4, nz))[:, None, None]
        frc.append(1e-9 * prof * base[None])
    frc = np.stack(frc)
    depth = 500.0 - 350.0 * np.clip(_modes(rng, lat, lon, 8, 4, 4), 0, None)
    frc[:, lev[:, None, None] > depth[None]] = np.nan
    F = Field(frc, ('time', 'lev', 'lat', 'lon'), {'lev': lev, 'lat': lat, 'lon': lon})
    iP = apps._update(apps.default_iParams, {'BCs': ['fixed', 'fixed', 'periodic']})
    mP = dict(apps.default_mParams); mP.update({'epsilon': 1e-5, 'k': 1e-7, 'N2': 1e-4 * (1.0 + np.linspace(0, 1, nz))})
    Fm, initS, cs = apps._coeffs_3DOcean(F, ['lev', 'lat', 'lon'], 'lat-lon', mP, iP, None)
    p3 = apps._cal_params3D(lev, lat, lon, 'lat-lon')
    return dict(kind='gen3d', zc=nz, yc=ny, xc=nx, BCz='fixed', BCy='fixed', BCx='periodic',
                delz=p3['del3'], dely=p3['del2'], delx=p3['del1'], delxSqr=p3['del1Sqr'],
                ratio2=p3['ratio2'], ratio1=p3['ratio1'], ratio2Sqr=p3['ratio2Sqr'],
                ratio1Sqr=p3['ratio1Sqr'], optArg=p3['optArg'], undef=apps._undeftmp, S0=initS.values,
                coefs=[np.ascontiguousarray(c) for c in cs] + [Fm.values], shared=(0, 1, 2, 3, 4, 5, 6))


def member(p, m):
    """Problem dict of one member (views) in the layout tests/util.py's runners take."""
    q = dict(p)
    q['S0'] = p['S0'][m]
    q['coefs'] = [c if k in p['shared'] else c[m] for k, c in enumerate(p['coefs'])]
    return q
