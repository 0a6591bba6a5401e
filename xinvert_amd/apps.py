"""Application front end: invert_Poisson / invert_Stommel / invert_StommelMunk /
invert_GillMatsuno / invert_omega.

Host-side mirror of the reference's callers of the SOR hot path (reference
xinvert/apps.py:67-100, 443-488, 535-582, 351-394, 766-827) with the same names, argument meaning,
defaults and error behaviour, restated on numpy (`Field`) because xarray is not available in
the build/test image.  xarray.DataArray inputs are accepted and returned when xarray exists.

The call shape is the reference's:  invert_*() -> _template() -> _coeffs_*() ->
_cal_params{2,3}D() -> core.inv_*() -> HIP kernels (xinvert_amd/csrc) through the C-ABI.
No arithmetic here runs on the GPU and none of it falls back to a CPU solver: core.inv_*
raises if the HIP library is missing.

Differences from the reference that a user can observe (all documented in DESIGN.md):
  * the solver is fp64; float32 inputs/coordinates are promoted to float64 on entry
    (the reference's mixed f32/f64 rounding is not reproduced);
  * sweeps are red-black / 4-colour ordered, so iterates (and loop counts) differ from the
    reference's lexicographic sweep while converged fields agree to <= 1e-6 rel-L2;
  * `flags` are reported per slice (iParams['flags'] has shape [nslice, 3]).
"""
import copy

import numpy as np

from . import core
from .field import Field, LazyForcing, aligned, along, from_any, full, to_like, undef_as

# default undefined value (reference apps.py:18, core.py:15)
_undeftmp = -9.99e8

# reference apps.py:21-38
default_iParams = copy.deepcopy({
    'BCs'      : ['fixed', 'fixed'],
    'undef'    : np.nan,
    'mxLoop'   : 5000,
    'tolerance': 1e-8,
    'optArg'   : None,
    'printInfo': True,
    'debug'    : False,
})

# reference apps.py:42-60
default_mParams = copy.deepcopy({
    'f0'     : 1e-5,
    'beta'   : 2e-11,
    'Phi'    : 1e4,
    'epsilon': 7e-6,
    'N2'     : 2e-4,
    'A'      : 1e5,
    'R'      : 5e-5,
    'depth'  : 100,
    'rho0'   : 1027,
    'ang0'   : 2e5,
    'lambda' : 1e-8,
    'c0'     : 8e-9,
    'c1'     : 8e-5,
    'Rearth' : 6371200.0,
    'Omega'  : 7.292e-5,
    'g'      : 9.80665,
})


# --------------------------------------------------------------------- entry points
def invert_Poisson(F, dims, coords='lat-lon', icbc=None,
                   mParams=default_mParams, iParams=default_iParams):
    """psi from F:  d2psi/dy2 + d2psi/dx2 = F   (reference apps.py:67-100)."""
    return _template(_coeffs_Poisson, core.inv_standard2D, 2, F, dims, coords,
                     icbc, ['g', 'Omega', 'Rearth'], mParams, iParams)


def invert_Stommel(curl, dims, coords='lat-lon', icbc=None,
                   mParams=default_mParams, iParams=default_iParams):
    """Stommel wind-driven gyre, general 2-D form (reference apps.py:443-488)."""
    return _template(_coeffs_Stommel, core.inv_general2D, 2, curl, dims, coords,
                     icbc, ['beta', 'R', 'D', 'rho0', 'g', 'Omega', 'Rearth'], mParams, iParams)


def invert_StommelMunk(curl, dims, coords='lat-lon', icbc=None,
                       mParams=default_mParams, iParams=default_iParams):
    """Stommel-Munk wind-driven gyre, biharmonic form (reference apps.py:535-582)."""
    return _template(_coeffs_StommelMunk, core.inv_general2D_bih, 2, curl, dims, coords,
                     icbc, ['A4', 'beta', 'R', 'D', 'rho0', 'g', 'Omega', 'Rearth'], mParams, iParams)


def invert_Fofonoff(F, dims, coords='cartesian', icbc=None,
                    mParams=default_mParams, iParams=default_iParams):
    """Fofonoff inertial gyre, standard 2-D "test" form (reference apps.py:721-763)."""
    return _template(_coeffs_Fofonoff, core.inv_standard2D_test, 2, F, dims, coords,
                     icbc, ['c0', 'c1', 'f0', 'beta', 'g', 'Omega', 'Rearth'], mParams, iParams)


def invert_BrethertonHaidvogel(h, dims, coords='cartesian', icbc=None,
                               mParams=default_mParams, iParams=default_iParams):
    """Bretherton-Haidvogel minimum-enstrophy flow over topography (reference apps.py:676-718)."""
    return _template(_coeffs_Bretherton, core.inv_standard2D_test, 2, h, dims, coords,
                     icbc, ['f0', 'beta', 'D', 'lambda', 'g', 'Omega', 'Rearth'], mParams, iParams)


def invert_GillMatsuno(Q, dims, coords='lat-lon', icbc=None,
                       mParams=default_mParams, iParams=default_iParams):
    """Gill-Matsuno mass field phi from heating Q (reference apps.py:351-394)."""
    return _template(_coeffs_GillMatsuno, core.inv_general2D, 2, Q, dims, coords,
                     icbc, ['f0', 'beta', 'epsilon', 'Phi', 'g', 'Omega', 'Rearth'],
                     mParams, iParams)


def invert_RefState(PV, dims, coords='z-lat', icbc=None,
                    mParams=default_mParams, iParams=default_iParams):
    """Balanced symmetric vortex: angular momentum from PV (reference apps.py:104-145).  As in the
    reference the accepted mParams key is 'Ang0' while the coefficients read the default 'ang0';
    'Gamma' has no default and must be supplied."""
    return _template(_coeffs_RefState, core.inv_standard2D, 2, PV, dims, coords, icbc,
                     ['Ang0', 'Gamma', 'g', 'Omega', 'Rearth'], mParams, iParams)


def invert_PV2D(PV, dims, coords='z-lat', icbc=None,
                mParams=default_mParams, iParams=default_iParams):
    """QG PV inversion in a vertical plane (reference apps.py:246-297)."""
    return _template(_coeffs_PV2D, core.inv_standard2D, 2, PV, dims, coords, icbc,
                     ['f0', 'beta', 'N2', 'g', 'Omega', 'Rearth'], mParams, iParams)


def invert_Eliassen(F, dims, coords='z-lat', icbc=None,
                    mParams=default_mParams, iParams=default_iParams):
    """Eliassen balanced-vortex model, 9-point standard form with user-supplied A, B, C
    (reference apps.py:300-346)."""
    return _template(_coeffs_Eliassen, core.inv_standard2D, 2, F, dims, coords, icbc,
                     ['A', 'B', 'C', 'g', 'Omega', 'Rearth'], mParams, iParams)


def invert_GillMatsuno_test(Q, dims, coords='lat-lon', icbc=None,
                            mParams=default_mParams, iParams=default_iParams):
    """Gill-Matsuno model in flux form (reference apps.py:397-442)."""
    return _template(_coeffs_GillMatsuno_test, core.inv_standard2D_test, 2, Q, dims, coords, icbc,
                     ['f0', 'beta', 'epsilon', 'Phi', 'g', 'Omega', 'Rearth'], mParams, iParams)


def invert_Stommel_test(curl, dims, coords='lat-lon', icbc=None,
                        mParams=default_mParams, iParams=default_iParams):
    """Stommel model in flux form (reference apps.py:491-534)."""
    return _template(_coeffs_Stommel_test, core.inv_standard2D_test, 2, curl, dims, coords, icbc,
                     ['beta', 'R', 'D', 'rho0', 'g', 'Omega', 'Rearth'], mParams, iParams)


def invert_StommelArons(Q, dims, coords='lat-lon', icbc=None,
                        mParams=default_mParams, iParams=default_iParams):
    """Stommel-Arons abyssal circulation (reference apps.py:585-629)."""
    return _template(_coeffs_StommelArons, core.inv_general2D, 2, Q, dims, coords, icbc,
                     ['f0', 'beta', 'epsilon', 'g', 'Omega', 'Rearth'], mParams, iParams)


def invert_geostrophic(lapPhi, dims, coords='lat-lon', icbc=None,
                       mParams=default_mParams, iParams=default_iParams):
    """Geostrophic streamfunction from the Laplacian of geopotential (reference apps.py:632-673)."""
    return _template(_coeffs_geostrophic, core.inv_standard2D, 2, lapPhi, dims, coords, icbc,
                     ['f0', 'beta', 'Omega', 'g', 'Omega', 'Rearth'], mParams, iParams)


def _check_N2(mParams):
    """Stratification profile sanity checks shared by the 3-D apps (reference apps.py:817-823,
    877-883): only array-valued N2 is checked, from its second level on."""
    N2 = mParams['N2'] if 'N2' in mParams else None
    if N2 is not None and not np.isscalar(N2):
        n2 = np.asarray(N2.values if hasattr(N2, 'values') else N2)
        if n2.ndim >= 1 and n2.shape[0] > 1:
            tail = n2[1:]
            if not np.isfinite(tail).all():
                raise Exception('inifinite stratification coefficient A')
            if np.isnan(tail).any():
                raise Exception('nan in coefficient A')
            if (tail <= 0).any():
                raise Exception('unstable stratification in coefficient A')


def invert_omega(F, dims, coords='lat-lon', icbc=None,
                 mParams=default_mParams, iParams=default_iParams):
    """QG omega equation, standard 3-D form (reference apps.py:766-827)."""
    _check_N2(mParams)
    return _template(_coeffs_omega, core.inv_standard3D, 3, F, dims, coords,
                     icbc, ['f0', 'beta', 'N2', 'g', 'Omega', 'Rearth'], mParams, iParams)


def invert_3DOcean(F, dims, coords='lat-lon', icbc=None,
                   mParams=default_mParams, iParams=default_iParams):
    """3-D wind-driven ocean flow with linear damping, general 3-D form
    (reference apps.py:830-888).  mParams needs 'k' (buoyancy damping) besides the defaults."""
    _check_N2(mParams)
    return _template(_coeffs_3DOcean, core.inv_general3D, 3, F, dims, coords, icbc,
                     ['f0', 'beta', 'epsilon', 'N2', 'k', 'g', 'Omega', 'Rearth'], mParams, iParams)


_ANIMATE = {
    # app name -> (coefficient function name, core function name, valid mParams)   (apps.py:949-1006)
    'poisson': ('_coeffs_Poisson', 'inv_standard2D', ['g', 'Omega', 'Rearth']),
    'gillmatsuno': ('_coeffs_GillMatsuno', 'inv_general2D',
                    ['f0', 'beta', 'epsilon', 'Phi', 'g', 'Omega', 'Rearth']),
    'stommel': ('_coeffs_Stommel', 'inv_general2D', ['beta', 'R', 'D', 'rho0', 'g', 'Omega', 'Rearth']),
    'stommelmunk': ('_coeffs_StommelMunk', 'inv_general2D_bih',
                    ['A4', 'beta', 'R', 'D', 'rho0', 'g', 'Omega', 'Rearth']),
    'brethertonhaidvogel': ('_coeffs_Bretherton', 'inv_standard2D_test',
                            ['f0', 'beta', 'D', 'lambda', 'g', 'Omega', 'Rearth']),
    'fofonoff': ('_coeffs_Fofonoff', 'inv_standard2D_test',
                 ['c0', 'c1', 'f0', 'beta', 'g', 'Omega', 'Rearth']),
    'omega': ('_coeffs_omega', 'inv_standard3D', ['f0', 'beta', 'N2', 'g', 'Omega', 'Rearth']),
    'pv2d': ('_coeffs_PV2D', 'inv_standard2D', ['f0', 'beta', 'N2', 'g', 'Omega', 'Rearth']),
    'geostrophic': ('_coeffs_geostrophic', 'inv_standard2D', ['f0', 'beta', 'g', 'Omega', 'Rearth']),
    'eliassen': ('_coeffs_Eliassen', 'inv_standard2D', ['A', 'B', 'C', 'g', 'Omega', 'Rearth']),
    'refstate': ('_coeffs_RefState', 'inv_standard2D', ['Ang0', 'Gamma', 'g', 'Omega', 'Rearth']),
    '3docean': ('_coeffs_3DOcean', 'inv_general3D',
                ['f0', 'beta', 'N2', 'epsilon', 'k', 'g', 'Omega', 'Rearth']),
}


def animate_iteration(app_name, F, dims, coords='lat-lon', icbc=None,
                      mParams=default_mParams, iParams=default_iParams,
                      loop_per_frame=5, max_frames=30):
    """Result after every `loop_per_frame` (+1) sweeps, stacked along a new leading `iter` axis
    (reference apps.py:895-1058).  Relies on the kernels being in place and restartable: every
    frame continues from the previous frame's S.  ('stommel' is routed to the general 2-D form;
    the reference's table pairs it with the biharmonic wrapper, which cannot take its 7 arrays.)"""
    tmpl = F
    F = from_any(F)
    if len(F.dims) != len(dims):
        raise Exception('For 2D case, only 2D slice  F is allowed;\n' +
                        'For 3D case, only 3D volume F is allowed.')
    name = app_name.lower()
    if name not in _ANIMATE:
        raise Exception('unsupported problem: ' + name + ', should be one of:\n' +
                        '\n'.join(repr(k) for k in _ANIMATE))
    coef_name, inv_name, validMPs = _ANIMATE[name]
    coef_func, invt_func = globals()[coef_name], getattr(core, inv_name)
    iParams_in = iParams
    iParams = _update(default_iParams, iParams)
    mParams = _update(default_mParams, mParams, validMPs)
    if icbc is not None:
        icbc = from_any(icbc)
    maskF, initS, coeffs = coef_func(F, dims, coords, mParams, iParams, icbc)
    if len(dims) == 2:
        ps = _cal_params2D(maskF[dims[0]], maskF[dims[1]], coords, Rearth=mParams['Rearth'])
    elif len(dims) == 3:
        ps = _cal_params3D(maskF[dims[0]], maskF[dims[1]], maskF[dims[2]], coords,
                           Rearth=mParams['Rearth'])
    else:
        raise Exception('dimension length should be one of [2, 3]')
    iParams = _update(ps, iParams)
    iParams['mxLoop'] = loop_per_frame
    iParams['printInfo'] = False
    # the coefficient stack, the forcing and S stay in HBM for all frames (core.Resident: torch holds the device
    # memory): a frame is one restart of the kernels on the resident batch plus one download of S.  Without torch
    # the frames go through the host-pointer inv_* call, as the reference does (apps.py:1031-1044).
    frames, frame_flags = [], []
    try:
        import torch  # noqa: F401
        res = core.Resident(inv_name, coeffs, maskF, initS, dims, iParams)
    except ImportError:
        res = None
    if res is not None:
        # every frame queued behind the last on the device (xinv_plan_solve_frames_f64_dev: no host round trip between the
        # frames while every frame runs its whole budget); the snapshots stay in HBM and cross PCIe in blocks
        res.keep_frames(max_frames)
        fl = res.solve_frames(max_frames, loop_per_frame, float(iParams['tolerance']))
        frame_flags = [np.array(f if res.rp.nb > 1 else f[0], copy=True) for f in fl]
    else:
        for _ in range(max_frames):
            initS = invt_func(*coeffs, maskF, initS, dims, iParams)
            frames.append(np.array(initS.values, copy=True))
            frame_flags.append(np.array(iParams['flags'], copy=True))
    iParams['frame_flags'] = np.stack(frame_flags)       # flags of every frame (iParams['flags'] holds the last)
    if isinstance(iParams_in, dict) and iParams_in is not default_iParams:
        for k in ('flags', 'stats', 'frame_flags'):      # where the inv_* calls leave them for the caller
            if k in iParams:
                iParams_in[k] = iParams[k]
    out = res.frames() if res is not None else np.stack(frames)
    if icbc is None:
        out = np.where(maskF.values[None] != _undeftmp, out, iParams['undef'])
    coords_out = dict(maskF.coords)
    coords_out['iter'] = np.arange(loop_per_frame, loop_per_frame * (max_frames + 1), loop_per_frame)
    res = Field(out, ('iter',) + maskF.dims, coords_out, name='inverted')
    return to_like(res, tmpl) if not isinstance(tmpl, Field) else res


def _deriv_center(vals, axis, coord, BC, scale=1.0, fill=0.0):
    """Centred first derivative with one padded point per side (reference finitediffs.py:548-659,
    `padBCs` + `deriv(scheme='center')`): pad by the boundary condition, extrapolate the
    coordinate linearly, numpy.gradient (== xarray's differentiate), drop the padded points."""
    coord = np.asarray(coord, dtype=np.float64)
    pw = [(0, 0)] * vals.ndim
    pw[axis] = (1, 1)
    if BC == 'periodic':
        p = np.pad(vals, pw, mode='wrap')
    elif BC == 'fixed':
        p = np.pad(vals, pw, mode='constant', constant_values=fill)
    elif BC == 'extend':
        p = np.pad(vals, pw, mode='edge')
    elif BC == 'reflect':
        p = np.pad(vals, pw, mode='reflect')
    else:
        raise Exception('unsupported BC: ' + str((BC, BC)))
    c = np.concatenate(([coord[0] * 2 - coord[1]], coord, [coord[-1] * 2 - coord[-2]]))
    g = np.gradient(p, c, axis=axis)
    sl = [slice(None)] * vals.ndim
    sl[axis] = slice(1, -1)
    return g[tuple(sl)] / scale


def cal_flow(S, dims, coords='lat-lon', BCs=('fixed', 'fixed'), vtype='streamfunction',
             mParams=default_mParams):
    """Flow components from an inverted field (reference apps.py:1181-1317).

    vtype 'streamfunction' -> (u, v) = (-dS/dy, dS/dx); 'velocitypotential' -> (dS/dx, dS/dy),
    centred differences on the BC-padded field with the lat-lon metric (finitediffs.py:151-207);
    'GillMatsuno' -> winds of the Gill-Matsuno mass field (apps.py:1277-1317), which also runs on
    the device (`cal_flow_gm_device`)."""
    if vtype.lower() not in ['streamfunction', 'velocitypotential', 'gillmatsuno']:
        raise Exception('unsupported vtype: ' + vtype + ', should be one of:\n' +
                        "['streamfunction', 'velocitypotential', 'gillmatsuno']")
    if vtype != 'GillMatsuno':
        tmpl = S
        S = from_any(S)
        sf = vtype == 'streamfunction'
        vals = np.asarray(S.values, dtype=np.float64)
        a0, a1 = S.axis(dims[0]), S.axis(dims[1])
        c0 = np.asarray(S[dims[0]], dtype=np.float64)
        c1 = np.asarray(S[dims[1]], dtype=np.float64)
        Rearth = default_mParams['Rearth']            # FiniteDiff's default R
        deg2m = np.pi * Rearth / 180.0
        out = lambda a, b: (to_like(S.like(a, 'u'), tmpl), to_like(S.like(b, 'v'), tmpl))
        c = coords.lower()
        if c == 'lat-lon':
            cos = along(np.cos(np.deg2rad(c0)), S, dims[0])
            grdy = _deriv_center(vals, a0, c0, BCs[0], deg2m)
            grdx = _deriv_center(vals, a1, c1, BCs[1], deg2m * cos)
            return out(-grdy, grdx) if sf else out(grdx, grdy)
        if c == 'z-lat':
            cos = along(np.cos(np.deg2rad(c1)), S, dims[1])
            grdz = _deriv_center(vals, a0, c0, BCs[0]) / cos
            grdy = _deriv_center(vals, a1, c1, BCs[1], deg2m) / cos
            grdy = np.where(along(np.abs(c1) != 90, S, dims[1]), grdy, 0.0)
            return out(-grdz, grdy) if sf else out(grdy, grdz)
        if c == 'z-lon':
            grdz = _deriv_center(vals, a0, c0, BCs[0])
            grdx = _deriv_center(vals, a1, c1, BCs[1], deg2m)      # no latitude dim: cos == 1
            return out(grdz, -grdx) if sf else out(grdx, grdz)
        if c == 'cartesian':
            grdy = _deriv_center(vals, a0, c0, BCs[0])
            grdx = _deriv_center(vals, a1, c1, BCs[1])
            return out(-grdy, grdx) if sf else out(grdx, grdy)
        raise Exception('unsupported coords ' + coords + ', should be [lat-lon, z-lat, z-lon, cartesian]')
    tmpl = S
    S = from_any(S)
    mParams = _update(default_mParams, mParams,
                      ['f0', 'beta', 'epsilon', 'Phi', 'Omega', 'Rearth'])
    eps, f0, beta = mParams['epsilon'], mParams['f0'], mParams['beta']
    Omega, Rearth = mParams['Omega'], mParams['Rearth']
    vals = np.asarray(S.values, dtype=np.float64)
    ay, ax = S.axis(dims[0]), S.axis(dims[1])
    yv = np.asarray(S[dims[0]], dtype=np.float64)
    xv = np.asarray(S[dims[1]], dtype=np.float64)
    Sy = np.gradient(vals, yv, axis=ay)       # == xarray .differentiate (edge_order=1)
    Sx = np.gradient(vals, xv, axis=ax)
    if coords.lower() == 'lat-lon':
        lats = np.deg2rad(yv)
        cosLat = along(np.cos(lats), S, dims[0])
        f = 2.0 * Omega * np.sin(lats)
        deg2m = np.deg2rad(1.0) * Rearth
        coef1 = along(eps / (eps**2.0 + f**2.0), S, dims[0])
        coef2 = along(f / (eps**2.0 + f**2.0), S, dims[0])
        c1 = - coef1 * Sx / deg2m / cosLat - coef2 * Sy / deg2m
        c2 = - coef1 * Sy / deg2m + coef2 * Sx / deg2m / cosLat
    elif coords.lower() == 'cartesian':
        f = f0 + beta * yv
        coef1 = along(eps / (eps**2.0 + f**2.0), S, dims[0])
        coef2 = along(f / (eps**2.0 + f**2.0), S, dims[0])
        c1 = - coef1 * Sx - coef2 * Sy
        c2 = - coef1 * Sy + coef2 * Sx
    else:
        raise Exception('unsupported coords ' + coords + ', should be [lat-lon, cartesian]')
    return to_like(S.like(c1, 'u'), tmpl), to_like(S.like(c2, 'v'), tmpl)


def gradient_tables(coord):
    """What numpy.gradient(f, coord, edge_order=1) (= xarray .differentiate) needs per index:
    (table[3n+3], uniform) with the non-uniform interior weights a, b, c and {dx, dx_first, dx_last}."""
    x = np.asarray(coord, dtype=np.float64)
    n = x.size
    d = np.diff(x)
    uniform = bool((d == d[0]).all())
    a = np.zeros(n); b = np.zeros(n); c = np.zeros(n)
    if n > 2:
        dx1, dx2 = d[:-1], d[1:]
        a[1:-1] = -(dx2) / (dx1 * (dx1 + dx2))
        b[1:-1] = (dx2 - dx1) / (dx1 * dx2)
        c[1:-1] = dx1 / (dx2 * (dx1 + dx2))
    tail = np.array([d[0], d[0], d[-1]])
    return np.concatenate([a, b, c, tail]), uniform


def cal_flow_gm_device(S_dev_ptr, u_dev_ptr, v_dev_ptr, nbatch, lat_or_y, lon_or_x, coords='lat-lon',
                       mParams=default_mParams, stream=None):
    """cal_flow(vtype='GillMatsuno') on fields that are already in HBM (device addresses of
    [nbatch, ny, nx] float64 arrays): the third field pair of config 4 without a host round trip.
    Bitwise equal to `cal_flow` on the same data.  Needs torch for the small coefficient tables."""
    import ctypes
    import torch
    from . import _lib
    L = _lib.require_gpu()
    mParams = _update(default_mParams, mParams, ['f0', 'beta', 'epsilon', 'Phi', 'Omega', 'Rearth'])
    eps, f0, beta = mParams['epsilon'], mParams['f0'], mParams['beta']
    Omega, Rearth = mParams['Omega'], mParams['Rearth']
    yv = np.asarray(lat_or_y, dtype=np.float64); xv = np.asarray(lon_or_x, dtype=np.float64)
    ytab, yuni = gradient_tables(yv)
    xtab, xuni = gradient_tables(xv)
    latlon = coords.lower() == 'lat-lon'
    if latlon:
        lats = np.deg2rad(yv)
        f = 2.0 * Omega * np.sin(lats)
        cosl = np.cos(lats)
        deg2m = np.deg2rad(1.0) * Rearth
    elif coords.lower() == 'cartesian':
        f = f0 + beta * yv
        cosl = np.ones_like(yv)
        deg2m = 1.0
    else:
        raise Exception('unsupported coords ' + coords + ', should be [lat-lon, cartesian]')
    rowtab = np.concatenate([eps / (eps**2.0 + f**2.0), f / (eps**2.0 + f**2.0), cosl])
    dev = torch.device('cuda', torch.cuda.current_device())
    ty, tx, tr = (torch.from_numpy(t).to(dev) for t in (ytab, xtab, rowtab))
    vp = ctypes.c_void_p
    sp = vp(stream) if stream else None
    rc = L.xinv_gm_flow_f64_dev(vp(S_dev_ptr), vp(u_dev_ptr), vp(v_dev_ptr), int(nbatch), yv.size, xv.size,
                                vp(ty.data_ptr()), vp(tx.data_ptr()), int(yuni), int(xuni),
                                vp(tr.data_ptr()), float(deg2m), int(latlon), sp)
    _lib.check(rc)


# ------------------------------------------------------------------------- helpers
def _template(coef_func, inv_func, dimLen, F, dims, coords='lat-lon', icbc=None,
              validParams=(), mParams=default_mParams, iParams=default_iParams):
    """Whole inverting process (reference apps.py:1324-1394)."""
    if len(dims) != dimLen:
        raise Exception('{0:2d} dimensional forcing are needed'.format(dimLen))

    tmpl = F
    F = from_any(F)
    if icbc is not None:
        icbc = from_any(icbc)

    iParams = _update(default_iParams, iParams)
    mParams = _update(default_mParams, mParams, list(validParams))

    # 1. coefficients (`_lazy`: builders that can may describe the forcing instead of computing it;
    #    iParams['device_prep'] = False keeps everything on the host, as the reference)
    iParams['_lazy'] = True
    maskF, initS, coeffs = coef_func(F, dims, coords, mParams, iParams, icbc)
    iParams.pop('_lazy', None)

    # 2. parameters
    if dimLen == 2:
        ps = _cal_params2D(maskF[dims[0]], maskF[dims[1]], coords, Rearth=mParams['Rearth'])
    elif dimLen == 3:
        ps = _cal_params3D(maskF[dims[0]], maskF[dims[1]], maskF[dims[2]], coords,
                           Rearth=mParams['Rearth'])
    else:
        raise Exception('dimension length should be one of [2, 3]')

    iParams = _update(ps, iParams)

    if iParams['debug']:
        print({k: v for k, v in iParams.items() if k != 'flags'})

    # 3. invert (HIP kernels; in place on initS -- or, with a lazy forcing, from zeros on the device
    #    with the forcing masked / scaled and the output de-masked there too)
    S = inv_func(*coeffs, maskF, initS, dims, iParams)

    # 4. de-mask
    if iParams.pop('_demasked', False):
        S = S.like(S.values, 'inverted')
    elif icbc is None:
        out = np.where(maskF.values != _undeftmp, S.values, iParams['undef'])
        S = S.like(out, 'inverted')
    else:
        S = S.like(S.values, 'inverted')
    S.iParams = iParams            # per-slice flags etc. for callers that want them
    return to_like(S, tmpl) if not isinstance(tmpl, Field) else S


def _mask_FS(F, dims, iParams, icbc, lazy=False):
    """Mask forcing with _undeftmp, build the initial guess (reference apps.py:2112-2159).

    lazy (builders whose forcing is `mask, then scale along one core dim`; no icbc): nothing is
    computed here -- the forcing comes back as a LazyForcing, the initial guess as None (zeros),
    and the solver's host entry does the masking / scaling / zero-fill / output de-mask on the
    device.  An infinity in the forcing then starts from S = 0 instead of the reference's
    `maskF - maskF` = NaN; both runs overflow at once."""
    if lazy and icbc is None and iParams.get('_lazy', False) and iParams.get('device_prep', True) \
            and np.asarray(F.values).dtype in (np.float64, np.float32):      # (float32 travels as float32: core._solve)
        return LazyForcing(F.values, F.dims, F.coords, iParams['undef'], _undeftmp, name=F.name), None, None
    vals = np.asarray(F.values, dtype=np.float64)
    # (a float32 forcing holds float32(undef): compare with the value its own dtype stores, as the reference's
    #  `F.where(F != undef)` does -- apps.py:2124-2128)
    undef = undef_as(np.asarray(F.values).dtype, iParams['undef'])
    if np.isnan(undef):
        mvals = np.where(np.isnan(vals), _undeftmp, vals)
    else:
        mvals = np.where(vals != undef, vals, _undeftmp)
    maskF = F.like(mvals)
    # `zero = maskF - maskF` of the reference: all zeros unless the forcing holds an infinity or (with
    # a value-undef) a NaN, which then turn into NaN.  Finite input is the rule: one cheap check
    # replaces two passes over the array and a copy.
    if np.isfinite(mvals).all():
        zero = np.zeros_like(mvals)
        zinit = np.zeros_like(mvals)
    else:
        zero = mvals - mvals
        zinit = zero.copy()

    if icbc is None:
        initS = F.like(zinit)
    else:
        mask = mvals == _undeftmp
        for dim, BC in zip(dims, iParams['BCs']):
            if BC != 'periodic':
                dv = np.asarray(F[dim])
                cond = along(np.isin(dv, [dv[0], dv[-1]]), F, dim)
                mask = np.logical_or(mask, cond)
        # aligned by dim NAME as xarray's `where` would (a labelled icbc whose dims are ordered
        # differently from F is transposed; an unknown dim raises)
        ic = np.broadcast_to(aligned(icbc, F), vals.shape)
        initS = F.like(np.where(mask, ic, 0.0))
    return maskF, initS, zero


def _remask(values, maskF):
    return np.where(maskF.values != _undeftmp, values, _undeftmp)


def _scale_remask(maskF, vec, dim):
    """`maskF * vec` along `dim`, masked points kept at _undeftmp (the builders' `F * cos(lat)` and
    re-mask) -- described, not computed, when the forcing is lazy."""
    if isinstance(maskF, LazyForcing):
        return maskF.scaled(vec, dim)
    return maskF.like(_remask(maskF.values * along(vec, maskF, dim), maskF))


def _as_forcing(maskF):
    return maskF if isinstance(maskF, LazyForcing) else maskF.like(maskF.values)


def _coeffs_Poisson(force, dims, coords, mParams, iParams, icbc):
    """reference apps.py:1397-1437.  Returns core-shaped (batch-shared) coefficients."""
    maskF, initS, zero = _mask_FS(force, dims, iParams, icbc, lazy=coords.lower() != 'z-lat')
    z2 = _core_zero(maskF, dims)
    c = coords.lower()
    if c == 'lat-lon':
        latd = np.asarray(maskF[dims[0]], dtype=np.float64)
        lats = np.deg2rad(latd)
        cosG = np.cos(lats)
        shifted = np.concatenate(([np.nan], lats[:-1]))
        cosH = np.cos((lats + shifted) / 2.0)
        A = z2 + cosH[:, None]
        B = _zero_like(z2)
        C = z2 + (1.0 / cosG)[:, None]
        Fv = _scale_remask(maskF, cosG, dims[0])
    elif c == 'z-lat':
        cosG = np.cos(np.deg2rad(np.asarray(maskF[dims[1]], dtype=np.float64)))
        A = z2 + 1.0
        B = _zero_like(z2)
        C = z2 + 1.0
        Fv = _scale_remask(maskF, cosG, dims[1])
    elif c in ('z-lon', 'cartesian'):
        A = z2 + 1.0
        B = _zero_like(z2)
        C = z2 + 1.0
        Fv = _as_forcing(maskF)
    else:
        raise Exception('unsupported coords ' + coords +
                        ', should be in [lat-lon, z-lat, z-lon, cartesian]')
    return Fv, initS, _cs(maskF, dims, A, B, C)


def _coeffs_Stommel(curl, dims, coords, mParams, iParams, icbc):
    """reference apps.py:1712-1748."""
    beta, R, depth = mParams['beta'], mParams['R'], mParams['D']
    rho0, Rearth, Omega = mParams['rho0'], mParams['Rearth'], mParams['Omega']
    maskF, initS, zero = _mask_FS(curl, dims, iParams, icbc)
    z2 = _core_zero(maskF, dims)
    R2 = _core_param(R, maskF, dims)          # scalar, or a field varying over the core dims
    c = coords.lower()
    if c == 'lat-lon':
        lats = np.deg2rad(np.asarray(curl[dims[0]], dtype=np.float64))
        cosL = np.cos(lats)
        A = z2 - R2 / depth
        B = _zero_like(z2)
        C = z2 - R2 / depth / (cosL**2.)[:, None]
        D = z2
        E = z2 - 2. * Omega / Rearth
        Fc = z2
    elif c == 'cartesian':
        A = z2 - R2 / depth
        B = _zero_like(z2)
        C = z2 - R2 / depth
        D = z2
        E = z2 - beta
        Fc = z2
    else:
        raise Exception('unsupported coords ' + coords +
                        ', should be in [lat-lon, z-lat, z-lon, cartesian]')
    G = _remask(-maskF.values / depth / rho0, maskF)
    return maskF.like(G), initS, _cs(maskF, dims, A, B, C, D, E, Fc)


def _coeffs_Fofonoff(f, dims, coords, mParams, iParams, icbc):
    """reference apps.py:1975-2013.  (The forcing argument only provides the grid and mask: the
    right-hand side is c1 - f(y), as in the reference.)"""
    f0, beta, c0, c1, Omega = (mParams[k] for k in ('f0', 'beta', 'c0', 'c1', 'Omega'))
    maskF, initS, zero = _mask_FS(f, dims, iParams, icbc)
    z2 = _core_zero(maskF, dims)
    yv = np.asarray(maskF[dims[0]], dtype=np.float64)
    c = coords.lower()
    if c == 'lat-lon':
        lats = np.deg2rad(yv)
        cosG = np.cos(lats)
        cosH = np.cos((lats + np.concatenate(([np.nan], lats[:-1]))) / 2.0)
        fc = 2. * Omega * np.sin(lats)
        A = z2 + cosH[:, None]
        B = _zero_like(z2)
        C = z2
        D = z2 + (1.0 / cosG)[:, None]
        E = z2 - (c0 * cosG)[:, None]
        Fv = _remask((zero + c1 - along(fc, maskF, dims[0])) * along(cosG, maskF, dims[0]), maskF)
    elif c == 'cartesian':
        fc = f0 + beta * yv
        A = z2 + 1.0
        B = _zero_like(z2)
        C = z2
        D = z2 + 1.0
        E = z2 - c0
        Fv = _remask(zero + c1 - along(fc, maskF, dims[0]), maskF)
    else:
        raise Exception('unsupported coords ' + coords + ', should be in [lat-lon, cartesian]')
    return maskF.like(Fv), initS, _cs(maskF, dims, A, B, C, D, E)


def _coeffs_Bretherton(h, dims, coords, mParams, iParams, icbc):
    """reference apps.py:1934-1972."""
    f0, beta, depth, lamb, Omega = (mParams[k] for k in ('f0', 'beta', 'D', 'lambda', 'Omega'))
    maskF, initS, zero = _mask_FS(h, dims, iParams, icbc)
    z2 = _core_zero(maskF, dims)
    yv = np.asarray(maskF[dims[0]], dtype=np.float64)
    c = coords.lower()
    if c == 'lat-lon':
        lats = np.deg2rad(yv)
        cosG = np.cos(lats)
        cosH = np.cos((lats + np.concatenate(([np.nan], lats[:-1]))) / 2.0)
        fc = 2. * Omega * np.sin(lats)
        A = z2 + cosH[:, None]
        B = _zero_like(z2)
        C = z2
        D = z2 + (1.0 / cosG)[:, None]
        E = z2 - (lamb * depth * cosG)[:, None]
        Fv = _remask(-maskF.values * along(fc, maskF, dims[0]) / depth * along(cosG, maskF, dims[0]), maskF)
    elif c == 'cartesian':
        fc = f0 + beta * yv
        A = z2 + 1.0
        B = _zero_like(z2)
        C = z2
        D = z2 + 1.0
        E = z2 - lamb * depth
        Fv = _remask(-maskF.values * along(fc, maskF, dims[0]) / depth, maskF)
    else:
        raise Exception('unsupported coords ' + coords + ', should be in [lat-lon, cartesian]')
    return maskF.like(Fv), initS, _cs(maskF, dims, A, B, C, D, E)


def _coeffs_StommelMunk(curl, dims, coords, mParams, iParams, icbc):
    """reference apps.py:1793-1836."""
    beta, A4, R, depth = mParams['beta'], mParams['A4'], mParams['R'], mParams['D']
    rho0, Omega, Rearth = mParams['rho0'], mParams['Omega'], mParams['Rearth']
    maskF, initS, zero = _mask_FS(curl, dims, iParams, icbc)
    z2 = _core_zero(maskF, dims)
    c = coords.lower()
    if c == 'lat-lon':
        lats = np.deg2rad(np.asarray(curl[dims[0]], dtype=np.float64))
        cosL = np.cos(lats)
        A = z2 + A4
        B = _zero_like(z2)
        C = z2 + (A4 / cosL**2.)[:, None]
        D = z2 - R / depth
        E = z2
        F = z2 - (R / depth / cosL**2.)[:, None]
        G = z2
        H = z2 - 2. * Omega / Rearth
        I = z2
    elif c == 'cartesian':
        A = z2 + A4
        B = _zero_like(z2)
        C = z2 + A4
        D = z2 - R / depth
        E = z2
        F = z2 - R / depth
        G = z2
        H = z2 - beta
        I = z2
    else:
        raise Exception('unsupported coords ' + coords + ', should be in [lat-lon, cartesian]')
    J = _remask(-maskF.values / depth / rho0, maskF)
    return maskF.like(J), initS, _cs(maskF, dims, A, B, C, D, E, F, G, H, I)


def _coeffs_GillMatsuno(Q, dims, coords, mParams, iParams, icbc):
    """reference apps.py:1609-1657."""
    Phi, epsilon = mParams['Phi'], mParams['epsilon']
    f0, beta = mParams['f0'], mParams['beta']
    Omega, Rearth = mParams['Omega'], mParams['Rearth']
    maskF, initS, zero = _mask_FS(Q, dims, iParams, icbc, lazy=True)
    z2 = _core_zero(maskF, dims)
    yv = np.asarray(Q[dims[0]], dtype=np.float64)
    c = coords.lower()
    if c == 'lat-lon':
        lats = np.deg2rad(yv)
        cosL = np.cos(lats)
        f = 2.0 * Omega * np.sin(lats)
        c1 = epsilon / (epsilon**2. + f**2.)
        c2 = f / (epsilon**2. + f**2.)
        deg2m = Rearth / 180. * np.pi
        A = z2 + (c1 * Phi)[:, None]
        B = _zero_like(z2)
        C = z2 + (c1 * Phi / cosL**2.)[:, None]
        D = z2 + (Phi * (np.gradient(c1, yv) / deg2m + c1 * np.tan(lats) / Rearth))[:, None]
        E = z2 - (Phi * np.gradient(c2, yv) / deg2m / cosL)[:, None]
        Fc = z2 - epsilon
    elif c == 'cartesian':
        f = f0 + beta * yv
        c1 = epsilon / (epsilon**2. + f**2.)
        c2 = f / (epsilon**2. + f**2.)
        A = z2 + (c1 * Phi)[:, None]
        B = _zero_like(z2)
        C = z2 + (c1 * Phi)[:, None]
        D = z2 + (Phi * np.gradient(c1, yv))[:, None]
        E = z2 - (Phi * np.gradient(c2, yv))[:, None]
        Fc = z2 - epsilon
    else:
        raise Exception('unsupported coords ' + coords + ', should be in [lat-lon, cartesian]')
    return _as_forcing(maskF), initS, _cs(maskF, dims, A, B, C, D, E, Fc)


def _coeffs_omega(force, dims, coords, mParams, iParams, icbc):
    """reference apps.py:2016-2052.  N2 may be a scalar or a profile over dims[0] (lev)."""
    f0, beta, N2, Omega = mParams['f0'], mParams['beta'], mParams['N2'], mParams['Omega']
    maskF, initS, zero = _mask_FS(force, dims, iParams, icbc, lazy=True)
    zc, yc, xc = (maskF.shape[maskF.axis(d)] for d in dims)
    z3 = np.zeros((zc, yc, 1))
    if np.isscalar(N2):
        n2 = N2
    else:
        n2 = np.asarray(N2.values if hasattr(N2, 'values') else N2, dtype=np.float64)
        if n2.ndim == 1:
            n2 = n2[:, None, None]
    yv = np.asarray(force[dims[1]], dtype=np.float64)
    c = coords.lower()
    if c == 'lat-lon':
        lats = np.deg2rad(yv)
        shifted = np.concatenate(([np.nan], lats[:-1]))
        cosH = np.cos((lats + shifted) / 2.)
        cosG = np.cos(lats)
        f = 2. * Omega * np.sin(lats)
        A = z3 + (f**2 * cosG)[None, :, None]
        B = z3 + n2 * cosH[None, :, None]
        C = z3 + n2 / cosG[None, :, None]
        Fv = _scale_remask(maskF, cosG, dims[1])
    elif c == 'cartesian':
        f = f0 + beta * yv
        A = z3 + (f**2.)[None, :, None]
        B = z3 + n2
        C = z3 + n2
        Fv = _as_forcing(maskF)
    else:
        raise Exception('unsupported coords ' + coords + ', should be in [lat-lon, cartesian]')
    return Fv, initS, _cs(maskF, dims, A, B, C)


def _half_shift(v):
    """(v + v.shift(1)) / 2 of the reference: half-point values, NaN in the first entry."""
    return (v + np.concatenate(([np.nan], v[:-1]))) / 2.


def _coeffs_RefState(Q, dims, coords, mParams, iParams, icbc):
    """reference apps.py:1440-1467.  C divides by the RAW input Q (not the masked copy) and, in
    'z-lat', by the latitude coordinate in degrees -- as written there."""
    ang0, Gamma, g = mParams['ang0'], mParams['Gamma'], mParams['g']
    maskF, initS, zero = _mask_FS(Q, dims, iParams, icbc)
    x1 = along(np.asarray(maskF[dims[1]], dtype=np.float64), maskF, dims[1])
    Qv = np.asarray(Q.values, dtype=np.float64)
    c = coords.lower()
    if c == 'z-lat':
        A = full(np.sin(np.deg2rad(x1)), maskF)
    elif c == 'cartesian':
        A = full(2.0 * ang0 / x1**3.0, maskF)
    else:
        raise Exception('unsupported coords ' + coords + ', should be in [z-lat, cartesian]')
    B = full(0.0, maskF)
    C = full(aligned(Gamma, maskF) * g / Qv / x1, maskF)
    return maskF.like(maskF.values), initS, _cs(maskF, dims, A, B, C)


def _coeffs_PV2D(PV, dims, coords, mParams, iParams, icbc):
    """reference apps.py:1556-1579 (both coordinate branches are the same expression)."""
    f0, N2 = mParams['f0'], mParams['N2']
    maskF, initS, zero = _mask_FS(PV, dims, iParams, icbc)
    if coords.lower() not in ('z-lat', 'cartesian'):
        raise Exception('unsupported coords ' + coords + ', should be in [z-lat, cartesian]')
    A = full(f0**2 / aligned(N2, maskF), maskF)
    B = full(0.0, maskF)
    C = full(1.0, maskF)
    return maskF.like(maskF.values), initS, _cs(maskF, dims, A, B, C)


def _coeffs_Eliassen(force, dims, coords, mParams, iParams, icbc):
    """reference apps.py:1582-1606: A, B, C are handed in by the caller (labelled arrays are
    aligned by dim name; NaNs in them are kept, as `zero + Am` keeps them)."""
    maskF, initS, zero = _mask_FS(force, dims, iParams, icbc)
    if coords.lower() not in ('z-lat', 'cartesian'):
        raise Exception('unsupported coords ' + coords + ', should be in [z-lat, cartesian]')
    A, B, C = (full(aligned(mParams[k], maskF), maskF) for k in ('A', 'B', 'C'))
    return maskF.like(maskF.values), initS, _cs(maskF, dims, A, B, C)


def _coeffs_GillMatsuno_test(Q, dims, coords, mParams, iParams, icbc):
    """reference apps.py:1660-1709: flux form, A on half points (NaN in row 0, never read)."""
    Phi, epsilon = mParams['Phi'], mParams['epsilon']
    f0, beta, Omega = mParams['f0'], mParams['beta'], mParams['Omega']
    maskF, initS, zero = _mask_FS(Q, dims, iParams, icbc)
    yv = np.asarray(Q[dims[0]], dtype=np.float64)
    col = lambda v: full(along(v, maskF, dims[0]), maskF)
    c = coords.lower()
    if c == 'lat-lon':
        lats = np.deg2rad(yv)
        cosG, cosH = np.cos(lats), np.cos(_half_shift(lats))
        fG = 2. * Omega * np.sin(lats)
        fH = 2. * Omega * np.sin(_half_shift(lats))
        c1G = epsilon / (epsilon**2. + fG**2.)
        c1H = epsilon / (epsilon**2. + fH**2.)
        c2G = fG / (epsilon**2. + fG**2.)
        A = col(c1H * Phi * cosH)
        B = col(0. - c2G * Phi)
        C = col(c2G * Phi)
        D = col(c1G * Phi / cosG)
        E = col(0. - epsilon * cosG)
        Fv = _remask(maskF.values * along(cosG, maskF, dims[0]), maskF)
    elif c == 'cartesian':
        fG = f0 + beta * yv
        fH = f0 + beta * _half_shift(yv)
        c1G = epsilon / (epsilon**2. + fG**2.)
        c1H = epsilon / (epsilon**2. + fH**2.)
        c2G = fG / (epsilon**2. + fG**2.)
        A = col(c1H * Phi)
        B = col(0. - c2G * Phi)
        C = col(c2G * Phi)
        D = col(c1G * Phi)
        E = full(0. - epsilon, maskF)
        Fv = maskF.values
    else:
        raise Exception('unsupported coords ' + coords + ', should be in [lat-lon, cartesian]')
    return maskF.like(Fv), initS, _cs(maskF, dims, A, B, C, D, E)


def _coeffs_Stommel_test(curl, dims, coords, mParams, iParams, icbc):
    """reference apps.py:1751-1790."""
    f0, beta, R, depth = mParams['f0'], mParams['beta'], mParams['R'], mParams['D']
    rho0, Omega = mParams['rho0'], mParams['Omega']
    maskF, initS, zero = _mask_FS(curl, dims, iParams, icbc)
    yv = np.asarray(curl[dims[0]], dtype=np.float64)
    col = lambda v: full(along(v, maskF, dims[0]), maskF)
    c = coords.lower()
    if c == 'lat-lon':
        lats = np.deg2rad(yv)
        cosG, cosH = np.cos(lats), np.cos(_half_shift(lats))
        f = 2. * Omega * np.sin(lats)
        A = col(0. - R / depth * cosH)
        B = col(0. - f)
        C = col(f)
        D = col(0. - R / depth / cosG)
        E = full(0.0, maskF)
        Fv = _remask(-maskF.values / depth / rho0 * along(cosG, maskF, dims[0]), maskF)
    elif c == 'cartesian':
        f = f0 + beta * yv
        A = full(0. - R / depth, maskF)
        B = col(0. - f)
        C = col(f)
        D = full(0. - R / depth, maskF)
        E = full(0.0, maskF)
        Fv = _remask(-maskF.values / depth / rho0, maskF)
    else:
        raise Exception('unsupported coords ' + coords + ', should be in [lat-lon, z-lat, z-lon, cartesian]')
    return maskF.like(Fv), initS, _cs(maskF, dims, A, B, C, D, E)


def _coeffs_StommelArons(Q, dims, coords, mParams, iParams, icbc):
    """reference apps.py:1839-1886 (the Gill-Matsuno operator with Phi = 1 and no damping term)."""
    epsilon, f0, beta = mParams['epsilon'], mParams['f0'], mParams['beta']
    Omega, Rearth = mParams['Omega'], mParams['Rearth']
    maskF, initS, zero = _mask_FS(Q, dims, iParams, icbc)
    yv = np.asarray(Q[dims[0]], dtype=np.float64)
    col = lambda v: full(along(v, maskF, dims[0]), maskF)
    c = coords.lower()
    if c == 'lat-lon':
        lats = np.deg2rad(yv)
        cosL = np.cos(lats)
        f = 2.0 * Omega * np.sin(lats)
        c1 = epsilon / (epsilon**2. + f**2.)
        c2 = f / (epsilon**2. + f**2.)
        deg2m = Rearth / 180. * np.pi
        A = col(c1)
        C = col(c1 / cosL**2.)
        D = col(np.gradient(c1, yv) / deg2m + c1 * np.tan(lats) / Rearth)
        E = col(0. - np.gradient(c2, yv) / deg2m / cosL)
    elif c == 'cartesian':
        f = f0 + beta * yv
        c1 = epsilon / (epsilon**2. + f**2.)
        c2 = f / (epsilon**2. + f**2.)
        A = col(c1)
        C = col(c1)
        D = col(np.gradient(c1, yv))
        E = col(0. - np.gradient(c2, yv))
    else:
        raise Exception('unsupported coords ' + coords + ', should be in [lat-lon, cartesian]')
    B = full(0.0, maskF)
    Fc = full(0.0, maskF)
    return _as_forcing(maskF), initS, _cs(maskF, dims, A, B, C, D, E, Fc)


def _coeffs_geostrophic(lapPhi, dims, coords, mParams, iParams, icbc):
    """reference apps.py:1889-1931.  The forcing is rebuilt from the RAW input (`lapPhi.where(
    lapPhi != undef)`), so NaN-masked input keeps its NaNs here exactly as in the reference;
    |f| < 2e-5 is inflated by 1.5 near the equator."""
    f0, beta, Omega = mParams['f0'], mParams['beta'], mParams['Omega']
    maskF, initS, zero = _mask_FS(lapPhi, dims, iParams, icbc)
    yv = np.asarray(maskF[dims[0]], dtype=np.float64)
    raw = np.asarray(lapPhi.values, dtype=np.float64)
    col = lambda v: full(along(v, maskF, dims[0]), maskF)
    c = coords.lower()
    if c == 'lat-lon':
        lats = np.deg2rad(yv)
        cosG, cosH = np.cos(lats), np.cos(_half_shift(lats))
        fH = 2. * Omega * np.sin(_half_shift(lats))
        fG = 2. * Omega * np.sin(lats)
        fH = np.where(np.abs(fH) < 2e-05, fH * 1.5, fH)
        fG = np.where(np.abs(fG) < 2e-05, fG * 1.5, fG)
        A = col(fH * cosH)
        C = col(fG / cosG)
        Fv = np.where(raw != _undeftmp, raw * along(cosG, maskF, dims[0]), _undeftmp)
    elif c == 'cartesian':
        fG = f0 + beta * yv
        fH = f0 + beta * _half_shift(yv)
        A = col(fH)
        C = col(fG)
        Fv = np.where(raw != _undeftmp, raw, _undeftmp)
    else:
        raise Exception('unsupported coords ' + coords + ', should be in [lat-lon, cartesian]')
    B = full(0.0, maskF)
    return maskF.like(Fv), initS, _cs(maskF, dims, A, B, C)


def _coeffs_3DOcean(force, dims, coords, mParams, iParams, icbc):
    """reference apps.py:2055-2109.  c3 = k / N2 (scalar or a profile over dims[0]); the
    derivative coefficients use numpy.gradient (== xarray's differentiate, edge_order 1) over the
    level and latitude coordinates; G is identically zero.  Every coefficient depends on
    (level, latitude) only, so they are built once on the core shape (batch stride 0)."""
    f0, beta, epsilon = mParams['f0'], mParams['beta'], mParams['epsilon']
    N2, k = mParams['N2'], mParams['k']
    Omega, Rearth = mParams['Omega'], mParams['Rearth']
    maskF, initS, zero = _mask_FS(force, dims, iParams, icbc)
    zc, yc, xc = (maskF.shape[maskF.axis(d)] for d in dims)
    z3 = np.zeros((zc, yc, 1))
    zv = np.asarray(force[dims[0]], dtype=np.float64)
    yv = np.asarray(force[dims[1]], dtype=np.float64)
    if np.isscalar(N2):
        c3 = (zv - zv) + k / N2
    else:
        c3 = (zv - zv) + k / np.asarray(N2.values if hasattr(N2, 'values') else N2, dtype=np.float64)
    lev = lambda v: v[:, None, None]
    lat = lambda v: v[None, :, None]
    c = coords.lower()
    if c == 'lat-lon':
        lats = np.deg2rad(yv)
        cosL = np.cos(lats)
        f = 2. * Omega * np.sin(lats)
        c1 = epsilon / (epsilon**2. + f**2.)
        c2 = f / (epsilon**2. + f**2.)
        deg2m = Rearth / 180. * np.pi
        A = z3 + lev(c3)
        B = z3 + lat(c1)
        C = z3 + lat(c1 / cosL**2.)
        D = z3 + lev(np.gradient(c3, zv))
        E = z3 + lat(np.gradient(c1, yv) / deg2m - c1 * np.tan(lats) / Rearth)
        Fc = z3 - lat(np.gradient(c2, yv) / deg2m / cosL)
        G = z3
    elif c == 'cartesian':
        f = f0 + beta * yv
        c1 = epsilon / (epsilon**2. + f**2.)
        c2 = f / (epsilon**2. + f**2.)
        A = z3 + lev(c3)
        B = z3 + lat(c1)
        C = z3 + lat(c1)
        D = z3 + lev(np.gradient(c3, zv))
        E = z3 + lat(np.gradient(c1, yv))
        Fc = z3 - lat(np.gradient(c2, yv))
        G = z3
    else:
        raise Exception('unsupported coords ' + coords + ', should be in [lat-lon, cartesian]')
    return maskF.like(maskF.values), initS, _cs(maskF, dims, A, B, C, D, E, Fc, G)


def _cs(maskF, dims, *coefs):
    """Coefficients at the core shape; the ones built on `_core_zero` become stride-0 views along x."""
    shape = tuple(maskF.shape[maskF.axis(d)] for d in dims)
    out = []
    for c in coefs:
        c = np.asarray(c)
        out.append(c if c.shape[-len(shape):] == shape else np.broadcast_to(c, shape))
    return tuple(out)


def _zero_like(z):
    """An identically-zero coefficient as a stride-0 view: core._solve recognises it and passes
    NULL for the cross coefficient B (nothing to pin, upload or scan on the device)."""
    return np.broadcast_to(np.float64(0.0), z.shape)


def _core_zero(maskF, dims):
    """`zero` restricted to the core dims, with the x axis left at length 1: coefficients that do
    not depend on the batch axis are built once and shared by every slice (batch stride 0 at the
    C-ABI), and those that depend on latitude / level only stay one value per row -- `_cs`
    broadcasts them to the core shape as stride-0 views, which core._solve ships as rows
    (xinv_options.rowconst_mask) -- instead of the reference's materialised broadcast to F's
    full shape (apps.py:2141)."""
    shape = tuple(maskF.shape[maskF.axis(d)] for d in dims)
    return np.zeros(shape[:-1] + (1,))


def _core_param(p, maskF, dims):
    if np.isscalar(p):
        return p
    p = np.asarray(p.values if hasattr(p, 'values') else p, dtype=np.float64)
    want = tuple(maskF.shape[maskF.axis(d)] for d in dims)
    if p.shape != want:
        raise Exception('field-valued parameter must have the core shape %r' % (want,))
    return p


def _cal_params2D(dim2_var, dim1_var, coords, Rearth=default_mParams['Rearth']):
    """reference apps.py:2245-2313."""
    dim2_var = np.asarray(dim2_var, dtype=np.float64)
    dim1_var = np.asarray(dim1_var, dtype=np.float64)
    gc2, gc1 = len(dim2_var), len(dim1_var)
    del2 = np.diff(dim2_var)[0]
    del1 = np.diff(dim1_var)[0]
    _uniform_interval(dim2_var, del2, 'dim2')
    _uniform_interval(dim1_var, del1, 'dim1')
    c = coords.lower()
    if c == 'lat-lon':
        del2 = np.deg2rad(del2) * Rearth
        del1 = np.deg2rad(del1) * Rearth
    elif c in ('z-lat', 'z-lon'):
        del1 = np.deg2rad(del1) * Rearth
    elif c == 'cartesian':
        pass
    else:
        raise Exception('unsupported coords for 2D case: ' + coords +
                        ', should be [lat-lon, cartesian]')
    ratio = del1 / del2
    epsilon = np.sin(np.pi/(2.0*gc1+2.0))**2 + np.sin(np.pi/(2.0*gc2+2.0))**2
    re = {
        'gc2': gc2, 'gc1': gc1, 'del2': del2, 'del1': del1,
        'ratio': ratio, 'ratioSSr': ratio ** 4.0, 'ratioSqr': ratio ** 2.0,
        'ratioQtr': ratio / 4.0, 'del1Sqr': del1 ** 2.0, 'del1Tr': del1 ** 3.0,
        'del1SSr': del1 ** 4.0,
        'optArg': 2.0 / (1.0 + np.sqrt((2.0 - epsilon) * epsilon)),
        'flags': np.array([0.0, 1.0, 0.0]),
    }
    return re


def _cal_params3D(dim3_var, dim2_var, dim1_var, coords, Rearth=default_mParams['Rearth']):
    """reference apps.py:2162-2242 (note the 2*gc3+3 in the z term, :2208)."""
    dim3_var = np.asarray(dim3_var, dtype=np.float64)
    dim2_var = np.asarray(dim2_var, dtype=np.float64)
    dim1_var = np.asarray(dim1_var, dtype=np.float64)
    gc3, gc2, gc1 = len(dim3_var), len(dim2_var), len(dim1_var)
    del3 = np.diff(dim3_var)[0]
    del2 = np.diff(dim2_var)[0]
    del1 = np.diff(dim1_var)[0]
    _uniform_interval(dim3_var, del3, 'dim3')
    _uniform_interval(dim2_var, del2, 'dim2')
    _uniform_interval(dim1_var, del1, 'dim1')
    c = coords.lower()
    if c == 'lat-lon':
        del2 = np.deg2rad(del2) * Rearth
        del1 = np.deg2rad(del1) * Rearth
    elif c == 'cartesian':
        pass
    else:
        raise Exception('unsupported coords for 3D case: ' + coords +
                        ', should be in [\'lat-lon\', \'cartesian\']')
    ratio1 = del1 / del2
    ratio2 = del1 / del3
    epsilon = (np.sin(np.pi/(2.0*gc1+2.0)) ** 2.0 +
               np.sin(np.pi/(2.0*gc2+2.0)) ** 2.0 +
               np.sin(np.pi/(2.0*gc3+3.0)) ** 2.0)
    re = {
        'gc3': gc3, 'gc2': gc2, 'gc1': gc1, 'del3': del3, 'del2': del2, 'del1': del1,
        'ratio1': ratio1, 'ratio2': ratio2,
        'ratio1Sqr': ratio1 ** 2.0, 'ratio2Sqr': ratio2 ** 2.0, 'del1Sqr': del1 ** 2.0,
        'optArg': 2.0 / (1.0 + np.sqrt((2.0 - epsilon) * epsilon)),
        'flags': np.array([0.0, 1.0, 0.0]),
    }
    return re


def _update(default, users, valid=None):
    """Merge user parameters over defaults (reference apps.py:2361-2375): None values are
    ignored; unknown mParams keys raise."""
    if valid is not None and users != default:
        for k in users:
            if k not in valid:
                raise Exception(f'mParams[\'{k}\'] is not used, valid are {valid}')
    out = copy.deepcopy(default)
    for k, v in users.items():
        if v is not None:
            out[k] = v
    return out


def _uniform_interval(coord1D, value, name='coordinate'):
    """reference apps.py:2377-2379."""
    if not np.isclose(np.diff(coord1D), value).all():
        raise Exception(f'coordinate {name} is non-uniform:\n{coord1D}')
