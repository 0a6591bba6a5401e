"""The register-resident solver for small slices (xinvert_amd/csrc/xinv_small2d.h, path 3): one
workgroup per slice, the whole solve in one launch.  Same red-black ordering as the streaming
kernels, so the checker is the same: the CPU oracle's coloured ordering, bit for bit, and the
streaming path on the same input."""
import numpy as np
import pytest

import util
from util import rand2d, run_oracle

pytestmark = pytest.mark.gpu
C2 = 2
PATH_FUSED, PATH_SMALL = 2, 3


def xuniform(kind, yc, xc, BCy, BCx, msk, seed, omega=1.3):
    """Random problem whose coefficient arrays are constant along x (what lat-lon builders produce)."""
    p = rand2d(kind, yc, xc, BCy, BCx, False, msk, seed=seed, omega=omega)
    n = 3 if kind == 'std2d' else 6
    for q in range(n):
        if q == 1:
            continue                                   # B == 0
        c = p['coefs'][q]
        c[:] = c[:, :1]
    return p


def small_rw(yc, xc, ext, gen=False):
    """The variant rule of xinv_hip.hip:small_variant -> (wavefronts, rows per wavefront), or None."""
    if xc > 384 or yc > 96 or (ext and xc % 2):
        return None
    nseg = -(-xc // 128)
    nr = 6 if gen else 4
    for nw in (16, 8):
        rws, cap = ((2, 4, 6), 12) if nw == 16 else ((4, 6, 8, 10, 12), 20)
        for rw in rws:
            if nw * rw < yc or rw * nseg > cap or (ext and (yc - 1) % rw == 0):
                continue
            if nseg == 3 and rw > (4 if nw == 16 else 6):
                continue
            lds = (nw + 2) * 2 * nseg * 2 * 64 * 8 + nw * rw * (nr * 8 + 4) + nw * 16 + 64 + yc * 2 * ((xc + 1) // 2) * 8
            if lds > 160 * 1024:
                continue
            return nw, rw
    return None


SHAPES = [(73, 144), (37, 50), (96, 384), (9, 16), (64, 300), (91, 130), (16, 128), (33, 256), (5, 8), (80, 129)]


@pytest.mark.parametrize('kind', ['std2d', 'gen2d'])
@pytest.mark.parametrize('BCy,BCx', [('fixed', 'fixed'), ('fixed', 'periodic'), ('extend', 'periodic'), ('extend', 'fixed')])
@pytest.mark.parametrize('shape', SHAPES)
def test_small_solver_vs_oracle(kind, BCy, BCx, shape):
    yc, xc = shape
    if BCx == 'periodic' and xc % 2:
        pytest.skip('odd xc periodic: seam colours, colour path')
    for msk in (0, 1):
        p = xuniform(kind, yc, xc, BCy, BCx, msk, seed=yc * 7 + xc + msk)
        So, flo = run_oracle(p, 30, 1e-9, C2)
        S, fl, st = util.run_hip_batched([p], 30, 1e-9, path=PATH_SMALL if small_rw(yc, xc, BCy == 'extend', kind == 'gen2d') else 0)
        applicable = small_rw(yc, xc, BCy == 'extend', kind == 'gen2d') is not None
        assert (st['path'] == PATH_SMALL) == applicable, (st, applicable)     # (path 3 is on request: not yet the engine's own choice)
        assert np.array_equal(S[0], So), 'mask %d: %d points differ' % (msk, (S[0] != So).sum())
        assert fl[0][2] == flo[2] and fl[0][0] == flo[0] and abs(fl[0][1] - flo[1]) <= 1e-12
        if applicable:
            S2, f2, s2 = util.run_hip_batched([p], 30, 1e-9, path=PATH_FUSED)
            assert s2['path'] == PATH_FUSED and np.array_equal(S, S2)


def test_small_solver_stops_at_the_exact_sweep_per_member():
    """Members converge after different numbers of sweeps; each returns the state of ITS stopping sweep."""
    ps = [xuniform('gen2d', 40, 96, 'fixed', 'periodic', 1, seed=100 + s, omega=1.3) for s in range(9)]
    S, fl, st = util.run_hip_batched(ps, 400, 3e-4, path=PATH_SMALL)
    assert st['path'] == PATH_SMALL
    loops = set()
    for m, p in enumerate(ps):
        So, flo = run_oracle(p, 400, 3e-4, C2)
        assert np.array_equal(S[m], So) and fl[m][2] == flo[2] and abs(fl[m][1] - flo[1]) <= 1e-12
        loops.add(int(flo[2]))
    assert len(loops) > 3 and max(loops) < 400


def test_small_solver_real_reference_shapes():
    """The reference's own regime: Gill-Matsuno on 73 x 144 (tests/test_GillMatsuno.py) and lat-lon Poisson
    with ['extend', 'periodic'] (tests/test_Poisson.py), through the synthetic front-end builders."""
    from xinvert_amd import synthetic
    p = synthetic.gill_matsuno(73, 144, 5)
    qs = [synthetic.member(p, m) for m in range(5)]
    S, fl, st = util.run_hip_batched(qs, 600, 1e-5, shared=p['shared'], path=PATH_SMALL)
    assert st['path'] == PATH_SMALL and st['xuniform_mask'] == 31
    for m in (0, 4):
        So, flo = run_oracle(qs[m], 600, 1e-5, C2)
        assert np.array_equal(S[m], So) and fl[m][2] == flo[2]
    for BCs in (('extend', 'periodic'), ('fixed', 'periodic')):
        p = synthetic.poisson_latlon(72, 144, mask=True, BCs=BCs, members=3)
        qs = [synthetic.member(p, m) for m in range(3)]
        S, fl, st = util.run_hip_batched(qs, 199, 0.0, shared=p['shared'], path=PATH_SMALL)
        assert st['path'] == PATH_SMALL and st['xuniform_mask'] == 3
        for m in range(3):
            So, flo = run_oracle(qs[m], 199, 0.0, C2)
            assert np.array_equal(S[m], So) and fl[m][2] == flo[2] == 199


def test_small_solver_degenerate_and_overflow():
    p = xuniform('std2d', 20, 40, 'fixed', 'fixed', 0, seed=5)
    for mx in (0, 1):
        So, flo = run_oracle(p, mx, 0.0, C2)
        S, fl, st = util.run_hip_batched([p], mx, 0.0, path=PATH_SMALL)
        assert st['path'] == PATH_SMALL and np.array_equal(S[0], So) and np.allclose(fl[0], flo, rtol=0, atol=1e-12)
    q = dict(p); q['coefs'] = [c.copy() for c in p['coefs']]
    q['coefs'][3][7, 9] = np.nan                       # NaN forcing: participates (NaN != undef), poisons S
    So, flo = run_oracle(q, 50, 0.0, C2)
    S, fl, st = util.run_hip_batched([q], 50, 0.0, path=PATH_SMALL)
    assert st['path'] == PATH_SMALL and flo[0] == 1 and fl[0][0] == 1 and fl[0][2] == flo[2]
    z = dict(p); z['coefs'] = [c.copy() for c in p['coefs']]; z['S0'] = np.zeros_like(p['S0'])
    z['coefs'][3][:] = 0.0                             # zero forcing, zero guess: norm == 0 stops standard_2D
    So, flo = run_oracle(z, 50, 0.0, C2)
    S, fl, st = util.run_hip_batched([z], 50, 0.0, path=PATH_SMALL)
    assert st['path'] == PATH_SMALL and np.array_equal(S[0], So) and np.allclose(fl[0], flo, rtol=0, atol=1e-12)


def test_small_solver_large_batch_on_device_pointers():
    """More slices than CUs (several rounds of workgroups), device-resident, shared coefficients."""
    from xinvert_amd import synthetic
    p = synthetic.gill_matsuno(73, 144, 600)
    qs = [synthetic.member(p, m) for m in range(600)]
    S, fl, st = util.run_hip_dev(qs, 39, 0.0, shared=p['shared'], path=PATH_SMALL)
    assert st['path'] == PATH_SMALL and st['sweep_launches'] == 1
    for m in (0, 255, 256, 599):
        So, flo = run_oracle(qs[m], 39, 0.0, C2)
        assert np.array_equal(S[m], So) and fl[m][2] == 39


# ------------------------------------------------------------------ 3-D: two sweeps per pass
def xuniform3d(zc, yc, xc, BCx, msk, seed):
    from util import rand3d
    p = rand3d(zc, yc, xc, 'fixed', BCx, msk, seed=seed)
    for q in range(3):
        c = p['coefs'][q]
        c[:] = c[:, :, :1]                             # constant along x (lat-lon omega coefficients)
    return p


@pytest.mark.parametrize('shape', [(9, 20, 66), (12, 33, 130), (50, 40, 250), (7, 17, 24), (64, 18, 120), (21, 50, 241)])
@pytest.mark.parametrize('BCx', ['fixed', 'periodic'])
def test_fused3d_two_sweeps_per_pass_vs_oracle(shape, BCx):
    """k_pipe3d (x-uniform coefficients; the planner's choice, asked for explicitly here: sweeps_per_launch = 2): passes of two sweeps + a
    one-sweep tail; must equal the oracle's coloured ordering and the one-sweep-per-pass kernel bit for bit."""
    zc, yc, xc = shape
    if BCx == 'periodic' and xc % 2:
        pytest.skip('odd xc periodic: seam colours, colour path')
    for msk in (0, 1):
        p = xuniform3d(zc, yc, xc, BCx, msk, seed=zc + yc + xc + msk)
        for nsw in (7, 8):                              # odd: 3 passes + tail; even: 4 passes
            So, flo = run_oracle(p, nsw - 1, 0.0, C2)
            S, fl, st = util.run_hip_batched([p], nsw - 1, 0.0, sweeps_per_launch=2)
            assert st['path'] == PATH_FUSED and st['sweeps_per_launch'] == 2 and st['xuniform_mask'] == 7, st
            assert np.array_equal(S[0], So), '%d points differ' % (S[0] != So).sum()
            assert fl[0][2] == flo[2] and abs(fl[0][1] - flo[1]) <= 1e-12
        S1, f1, s1 = util.run_hip_batched([p], 7, 0.0, sweeps_per_launch=1)
        assert s1['sweeps_per_launch'] == 1 and np.array_equal(S1, S)


def test_fused3d_two_sweeps_tolerance_stop_inside_a_pass():
    """Members stop at different sweeps, some on the first sweep of a two-sweep pass (redo from the
    pass's source buffer)."""
    ps = [xuniform3d(10, 24, 64, 'periodic', 1, seed=40 + s) for s in range(6)]
    S, fl, st = util.run_hip_batched(ps, 300, 2e-3, sweeps_per_launch=2)
    assert st['sweeps_per_launch'] == 2
    par = set()
    for m, p in enumerate(ps):
        So, flo = run_oracle(p, 300, 2e-3, C2)
        assert np.array_equal(S[m], So) and fl[m][2] == flo[2], (m, fl[m], flo)
        par.add(int(flo[2]) % 2)
    assert par == {0, 1}


def test_fused3d_two_sweeps_k_chunks_tall_volume():
    """A tall, narrow volume is split into k chunks (four recomputed halo planes a side)."""
    p = xuniform3d(200, 20, 100, 'fixed', 1, seed=3)
    So, flo = run_oracle(p, 5, 0.0, C2)
    S, fl, st = util.run_hip_batched([p], 5, 0.0, sweeps_per_launch=2)
    assert st['sweeps_per_launch'] == 2 and np.array_equal(S[0], So) and fl[0][2] == flo[2]
