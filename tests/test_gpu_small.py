"""Small slices -- the reference's own regime, 73 x 144 fields and a few hundred of them -- on the engine's own
choice of kernel (the streaming kernels; the register-resident one-launch solver of rounds 2-3, path 3, was measured
2.7 x slower on exactly these shapes and removed in round 4), and the 3-D two-sweep pass.  The checker is the CPU
oracle's coloured ordering, bit for bit."""
import numpy as np
import pytest

import util
from util import rand2d, run_oracle

pytestmark = pytest.mark.gpu
C2 = 2
PATH_FUSED = 2


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


SHAPES = [(73, 144), (37, 50), (96, 384), (9, 16), (64, 300), (91, 130), (16, 128), (33, 256), (5, 8), (80, 129)]


@pytest.mark.parametrize('kind', ['std2d', 'gen2d'])
@pytest.mark.parametrize('BCy,BCx', [('fixed', 'fixed'), ('fixed', 'periodic'), ('extend', 'periodic'), ('extend', 'fixed')])
@pytest.mark.parametrize('shape', SHAPES)
def test_small_slices_vs_oracle(kind, BCy, BCx, shape):
    yc, xc = shape
    for msk in (0, 1):
        p = xuniform(kind, yc, xc, BCy, BCx, msk, seed=yc * 7 + xc + msk)
        So, flo = run_oracle(p, 30, 1e-9, C2)
        S, fl, st = util.run_hip_batched([p], 30, 1e-9)
        assert np.array_equal(S[0], So), 'mask %d: %d points differ' % (msk, (S[0] != So).sum())
        assert fl[0][2] == flo[2] and fl[0][0] == flo[0] and abs(fl[0][1] - flo[1]) <= 1e-12


def test_removed_small_path_is_refused():
    """xinv_options.path = 3 named the removed solver: a clear error, not a silent other kernel."""
    from xinvert_amd import _lib
    p = xuniform('std2d', 20, 40, 'fixed', 'fixed', 0, seed=5)
    with pytest.raises(_lib.XinvError, match='removed'):
        util.run_hip_batched([p], 5, 0.0, path=3)


def test_small_slices_stop_at_the_exact_sweep_per_member():
    """Members converge after different numbers of sweeps; each returns the state of ITS stopping sweep."""
    ps = [xuniform('gen2d', 40, 96, 'fixed', 'periodic', 1, seed=100 + s, omega=1.3) for s in range(9)]
    S, fl, st = util.run_hip_batched(ps, 400, 3e-4)
    loops = set()
    for m, p in enumerate(ps):
        So, flo = run_oracle(p, 400, 3e-4, C2)
        assert np.array_equal(S[m], So) and fl[m][2] == flo[2] and abs(fl[m][1] - flo[1]) <= 1e-12
        loops.add(int(flo[2]))
    assert len(loops) > 3 and max(loops) < 400


def test_small_slices_real_reference_shapes():
    """The reference's own regime: Gill-Matsuno on 73 x 144 (tests/test_GillMatsuno.py) and lat-lon Poisson
    with ['extend', 'periodic'] (tests/test_Poisson.py), through the synthetic front-end builders."""
    from xinvert_amd import synthetic
    p = synthetic.gill_matsuno(73, 144, 5)
    qs = [synthetic.member(p, m) for m in range(5)]
    S, fl, st = util.run_hip_batched(qs, 600, 1e-5, shared=p['shared'])
    assert st['path'] == PATH_FUSED and st['xuniform_mask'] == 31
    for m in (0, 4):
        So, flo = run_oracle(qs[m], 600, 1e-5, C2)
        assert np.array_equal(S[m], So) and fl[m][2] == flo[2]
    for BCs in (('extend', 'periodic'), ('fixed', 'periodic')):
        p = synthetic.poisson_latlon(72, 144, mask=True, BCs=BCs, members=3)
        qs = [synthetic.member(p, m) for m in range(3)]
        S, fl, st = util.run_hip_batched(qs, 199, 0.0, shared=p['shared'])
        assert st['path'] == PATH_FUSED and st['xuniform_mask'] == 3
        for m in range(3):
            So, flo = run_oracle(qs[m], 199, 0.0, C2)
            assert np.array_equal(S[m], So) and fl[m][2] == flo[2] == 199


def test_small_slices_large_batch_on_device_pointers():
    """More slices than CUs (several rounds of workgroups), device-resident, shared coefficients."""
    from xinvert_amd import synthetic
    p = synthetic.gill_matsuno(73, 144, 600)
    qs = [synthetic.member(p, m) for m in range(600)]
    S, fl, st = util.run_hip_dev(qs, 39, 0.0, shared=p['shared'])
    assert st['path'] == PATH_FUSED
    for m in (0, 255, 256, 599):
        So, flo = run_oracle(qs[m], 39, 0.0, C2)
        assert np.array_equal(S[m], So) and fl[m][2] == 39


# ------------------------------------------------------------------ 3-D: two sweeps per pass
def xuniform3d(zc, yc, xc, BCx, msk, seed, BCy='fixed'):
    from util import rand3d
    p = rand3d(zc, yc, xc, BCy, BCx, msk, seed=seed)
    for q in range(3):
        c = p['coefs'][q]
        c[:] = c[:, :, :1]                             # constant along x (lat-lon omega coefficients)
    return p


@pytest.mark.parametrize('shape', [(9, 20, 66), (12, 33, 130), (50, 40, 250), (7, 17, 24), (64, 18, 120), (21, 50, 241), (8, 39, 65),
                                   (6, 21, 121), (6, 22, 121), (6, 23, 121)])
@pytest.mark.parametrize('BCx', ['fixed', 'periodic'])
@pytest.mark.parametrize('BCy', ['fixed', 'extend'])
def test_fused3d_two_sweeps_per_pass_vs_oracle(shape, BCx, BCy):
    """k_pipe3d (x-uniform coefficients; the planner's choice, asked for explicitly here: sweeps_per_launch = 2): passes of two sweeps + a
    one-sweep tail; must equal the oracle's coloured ordering and the one-sweep-per-pass kernel bit for bit.  Round 5: odd
    widths with periodic x run the two-sweep pass too (the even-ring variant).  Round 6: BCy = 'extend' runs it as well (the
    first sweep's pre-pass by k_extend on the source, the second's inside the kernel: xinv_pipe3d.h EXT) where rows yc-2 / yc-1
    share a wavefront (util.p3_extend_ok: two row counts in three) -- not with the odd-xc periodic seam; the rest keep the
    one-sweep kernel."""
    zc, yc, xc = shape
    for msk in (0, 1):
        p = xuniform3d(zc, yc, xc, BCx, msk, seed=zc + yc + xc + msk, BCy=BCy)
        for nsw in (7, 8):                              # odd: 3 passes + tail; even: 4 passes
            So, flo = run_oracle(p, nsw - 1, 0.0, C2)
            S, fl, st = util.run_hip_batched([p], nsw - 1, 0.0, sweeps_per_launch=2)
            one = BCy == 'extend' and ((BCx == 'periodic' and xc % 2 == 1) or not util.p3_extend_ok(yc))
            assert st['path'] == PATH_FUSED and st['sweeps_per_launch'] == (1 if one else 2) and st['xuniform_mask'] == 7, st
            assert np.array_equal(S[0], So), '%d points differ' % (S[0] != So).sum()
            assert fl[0][2] == flo[2] and abs(fl[0][1] - flo[1]) <= 1e-12
        S1, f1, s1 = util.run_hip_batched([p], 7, 0.0, sweeps_per_launch=1)
        assert s1['sweeps_per_launch'] == 1 and np.array_equal(S1, S)


@pytest.mark.parametrize('BCy', ['fixed', 'extend'])
def test_fused3d_two_sweeps_tolerance_stop_inside_a_pass(BCy):
    """Members stop at different sweeps, some on the first sweep of a two-sweep pass (redo from the
    pass's source buffer -- with 'extend' that source carries the first sweep's pre-pass already: idempotent)."""
    ps = [xuniform3d(10, 24, 64, 'periodic', 1, seed=40 + s, BCy=BCy) for s in range(6 if BCy == 'fixed' else 14)]
    S, fl, st = util.run_hip_batched(ps, 300, 2e-3, sweeps_per_launch=2)
    assert st['sweeps_per_launch'] == 2
    par = set()
    for m, p in enumerate(ps):
        So, flo = run_oracle(p, 300, 2e-3, C2)
        assert np.array_equal(S[m], So) and fl[m][2] == flo[2], (m, fl[m], flo)
        par.add(int(flo[2]) % 2)
    assert par == {0, 1}


@pytest.mark.parametrize('BCy', ['fixed', 'extend'])
def test_fused3d_two_sweeps_k_chunks_tall_volume(BCy):
    """A tall, narrow volume is split into k chunks (four recomputed halo planes a side)."""
    p = xuniform3d(200, 20, 100, 'fixed', 1, seed=3, BCy=BCy)
    So, flo = run_oracle(p, 5, 0.0, C2)
    S, fl, st = util.run_hip_batched([p], 5, 0.0, sweeps_per_launch=2)
    assert st['sweeps_per_launch'] == 2 and np.array_equal(S[0], So) and fl[0][2] == flo[2]
