#!/usr/bin/env python3
"""Seeded fuzz of the resident plans (include/xinv.h, xinv_plan_*) on a GPU box: the medium generator of
tests/test_gpu_parity.py (every form, batches of two on one coefficient stack, masks, 'extend', x-uniform variants, odd
widths, tolerance stops) with every case solved on a PLAN -- built by the first solve, reused by a second solve from the
same initial state (must give the same bits), then continued from its own result for a few more sweeps (the reference's
restart idiom, apps.py:1031-1044) -- against the oracle, bit for bit.  x-uniform coefficient stacks travel as one value per
row (rowconst_mask) in every other case.

  python tests/fuzz_plan.py [first_chunk] [count] [--stops]      (--stops: every case a tolerance stop)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))

STATS = {'cases': 0, 'rows': 0, 'skip': 0, 'pq': 0, 'lag_flip': 0}


def plan_runner(ps, nsw, tol, shared=(), **opt):
    import util
    from oracle import COLOUR_AUTO
    from xinvert_amd.resident import ResidentProblem
    ncoef = len(ps[0]['coefs'])
    p = dict(ps[0])
    p['S0'] = np.stack([q['S0'] for q in ps])
    rows = STATS['cases'] & 1
    cs = []
    for k in range(ncoef):
        a = ps[0]['coefs'][k] if k in shared else np.stack([q['coefs'][k] for q in ps])
        if rows and k < ncoef - 1 and np.array_equal(a, np.broadcast_to(a[..., :1], a.shape)):
            a = np.broadcast_to(np.ascontiguousarray(a[..., :1]), a.shape)        # a stride-0 view along x: travels as rows
        cs.append(a)
    p['coefs'] = cs
    p['shared'] = tuple(shared)
    rp = ResidentProblem(p)
    STATS['cases'] += 1
    STATS['rows'] += rp.rowconst != 0
    fl, st = rp.solve(nsw, tol, **opt)
    S = rp.result(); fl = np.array(fl, copy=True)
    assert st['planned'] == 1
    STATS['skip'] += st['masked_tile_pct'] > 0
    STATS['pq'] += st['point_factor'] > 0
    rp.reset()
    fl2, st2 = rp.solve(nsw, tol, **opt)                  # the plan, reused: the same bits
    assert np.array_equal(rp.result(), S, equal_nan=True) and np.array_equal(fl2, fl, equal_nan=True), 'second solve on the plan differs'
    if not np.isnan(S).any():                             # continue from the result, as animate_iteration does frame after frame
        more = 1 + (STATS['cases'] % 5)
        fl3, _ = rp.solve(more, 0.0, **opt)
        S3 = rp.result()
        for m, q in enumerate(ps):
            q2 = dict(q); q2['S0'] = S[m]
            So, flo = util.run_oracle(q2, more, 0.0, COLOUR_AUTO)
            assert np.array_equal(S3[m], So, equal_nan=True) and fl3[m][2] == flo[2], 'continuation on the plan differs (member %d)' % m
    rp.close()
    return S, fl, st


def main():
    import test_gpu_parity as t
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    first = int(args[0]) if len(args) > 0 else 1000
    count = int(args[1]) if len(args) > 1 else 40
    stops = '--stops' in sys.argv
    bad = 0
    for chunk in range(first, first + count):
        for odd in (0, 1):
            try:
                t._medium_fuzz(chunk, 7000 if not odd else 9000, odd, every_case_stops=stops, runner=plan_runner)
            except Exception as e:
                bad += 1
                print('FAIL chunk', chunk, 'odd' if odd else 'even', str(e)[:500])
    print('plan fuzz: chunks %d..%d x {even, odd}%s: %d cases (%d with row-constant uploads, %d with skipped tiles, %d on the point-factor stream), failures: %d'
          % (first, first + count - 1, ', every case a tolerance stop' if stops else '', STATS['cases'], STATS['rows'], STATS['skip'], STATS['pq'], bad))
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
