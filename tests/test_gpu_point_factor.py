"""General 2-D form with coefficients that vary along x (Stommel with R(x, y): BASELINE configs[2]; reference
numbas.py:1125-1153) on the point-factor stream of round 5: the relaxation factor optArg / ((A ratioSqr + C) 2 - F delxSqr)
and the update predicate of every point are evaluated once per coefficient stack (k_point_factor) and read by the sweeps
as one more stream (FusedGen2DQ; FusedGen2DQA when A and C hold the same numbers).  Same expression, same bits: the runs
must equal the in-kernel evaluation (XINV_FLAG_NO_POINT_FACTOR) and the oracle, bit for bit."""
import numpy as np
import pytest

import util
from oracle import COLOUR_2

pytestmark = pytest.mark.gpu


def _mk(yc, xc, BCy, BCx, msk, seed, rows_def, alias):
    q = util.rand2d('gen2d', yc, xc, BCy, BCx, 0, msk, seed=seed)
    if alias:
        q['coefs'][2] = q['coefs'][0].copy()                  # C = A, masks included (an isotropic operator)
    if rows_def:
        for k in (3, 4, 5):                                   # D, E, F constant along x
            q['coefs'][k] = np.repeat(q['coefs'][k][:, :1], xc, axis=1)
    return q


@pytest.mark.parametrize('alias', [0, 1])
@pytest.mark.parametrize('rows_def', [0, 1])
@pytest.mark.parametrize('BCy,BCx', [('fixed', 'fixed'), ('fixed', 'periodic'), ('extend', 'periodic'), ('extend', 'fixed')])
@pytest.mark.parametrize('shape', [(48, 140), (131, 520), (300, 1026)])
def test_point_factor_stream_equals_in_kernel_factor_and_oracle(shape, BCy, BCx, rows_def, alias):
    yc, xc = shape
    ps = [_mk(yc, xc, BCy, BCx, m & 1, 40 + 3 * m, rows_def, alias) for m in range(3)]
    want_um = (28 if rows_def else 0)
    for spl in (0, 1, 2):
        for mx, tol in ((13, 0.0), (400, 2e-3)):
            S, fl, st = util.run_hip_dev(ps, mx, tol, sweeps_per_launch=spl)
            S1, fl1, st1 = util.run_hip_dev(ps, mx, tol, sweeps_per_launch=spl, no_point_factor=1)
            assert st['path'] == 2 and st['xuniform_mask'] == want_um and st['pipelined'] == 0, st
            assert st['point_factor'] == (2 if alias else 1) and st1['point_factor'] == 0, (st['point_factor'], st1['point_factor'])
            assert np.array_equal(S, S1) and np.array_equal(fl, fl1), (shape, BCy, BCx, rows_def, alias, spl, mx)
            for m, q in enumerate(ps):
                So, flo = util.run_oracle(q, mx, tol, COLOUR_2)
                assert np.array_equal(S[m], So) and fl[m][2] == flo[2], (shape, BCy, BCx, rows_def, alias, spl, mx, m)


def test_point_factor_with_masked_tiles_and_a_resident_plan():
    """Tile lists (whole tiles of undefined forcing) + a plan that keeps Q: repeated solves, then new coefficients and a refresh."""
    import torch
    from xinvert_amd.resident import ResidentProblem
    q = _mk(420, 1300, 'fixed', 'fixed', 1, 77, 1, 1)
    q['coefs'][-1][30:260, 200:900] = q['undef']
    p = dict(q); p['S0'] = q['S0'][None]; p['coefs'] = [c[None] for c in q['coefs']]; p['shared'] = ()
    rp = ResidentProblem(p)
    ref = util.run_oracle(q, 25, 0.0, COLOUR_2)[0]
    for _ in range(3):
        rp.reset()
        fl, st = rp.solve(25, 0.0, force_tile_skip=1)
        assert st['planned'] == 1 and st['masked_tile_pct'] > 0
        assert np.array_equal(rp.result()[0], ref)
    q2 = _mk(420, 1300, 'fixed', 'fixed', 1, 78, 1, 0)       # A != C now: the alias variant must go
    for k, c in enumerate(q2['coefs']):
        rp.coefs[k].copy_(torch.from_numpy(np.ascontiguousarray(c)[None]).to(rp.dev))
    rp.S0.copy_(torch.from_numpy(q2['S0'][None]).to(rp.dev)); rp.reset()
    rp.refresh()
    rp.solve(25, 0.0, force_tile_skip=1)
    assert np.array_equal(rp.result()[0], util.run_oracle(q2, 25, 0.0, COLOUR_2)[0])
    rp.close()


def test_zero_relaxation_factor_falls_back_to_the_in_kernel_evaluation():
    """Q == 0 means "skip this point": a stack on which a true factor is +-0 (an infinite denominator) cannot use the
    stream, and the engine evaluates in the kernel as before -- same result as asking for that explicitly."""
    q = _mk(60, 200, 'fixed', 'fixed', 0, 5, 1, 0)
    q['coefs'][0][20, 50] = np.inf                            # A = inf: optArg / inf = 0 at that point
    S, fl, st = util.run_hip_dev([q], 6, 0.0)
    S1, fl1, st1 = util.run_hip_dev([q], 6, 0.0, no_point_factor=1)
    assert st['point_factor'] == 0                            # (refused: a zero factor on an updatable point)
    assert np.array_equal(S, S1, equal_nan=True) and np.array_equal(fl, fl1, equal_nan=True)
    # (the infinity poisons S -- the overflow exit -- and how far a NaN has spread when the run stops is the one thing the
    #  B == 0 kernels do not share with the oracle, DESIGN.md 2: the exit sweep and the flags are the same)
    So, flo = util.run_oracle(q, 6, 0.0, COLOUR_2)
    assert fl[0][0] == flo[0] == 1.0 and fl[0][2] == flo[2]
