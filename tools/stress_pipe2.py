"""Replays the test pair around the one intermittent mismatch seen in the suite (k_pipe2d bring-up)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np
from util import rand2d, run_oracle, run_hip_batched
from test_gpu_parity import _uniform2d, _seed

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
def mk(BCy, BCx, msk, shape):
    p = _uniform2d(rand2d('std2d', shape[0], shape[1], BCy, BCx, 0, msk, seed=_seed(('k34', BCy, BCx, msk, shape))))
    So, flo = run_oracle(p, 24, 1e-9, 2)
    return p, So, flo
prev = [mk('extend', 'fixed', 1, (70, 130)), mk('extend', 'periodic', 1, (70, 130)), mk('fixed', 'extend', 1, (70, 130))]
cur = mk('fixed', 'fixed', 0, (33, 257))
bad = 0
for it in range(n):
    for p, So, flo in prev:
        for rows in (16, 0):
            S, fl, st = run_hip_batched([p], 24, 1e-9, path=2, sweeps_per_launch=4, rows_per_tile=rows)
            if not np.array_equal(S[0], So): print('prev mismatch', it, rows, flush=True); bad += 1
        S, fl, st = run_hip_batched([p], 24, 1e-9, path=2)
    p, So, flo = cur
    for rows in (16, 0):
        S, fl, st = run_hip_batched([p], 24, 1e-9, path=2, sweeps_per_launch=4, rows_per_tile=rows)
        if not np.array_equal(S[0], So):
            bad += 1
            d = (S[0] != So)
            rb = np.where(d.any(axis=1))[0]; cb = np.where(d.any(axis=0))[0]
            print('iter', it, 'rows', rows, 'points', int(d.sum()), 'rows', rb.min(), rb.max(), len(rb), 'cols', cb.min(), cb.max(), len(cb), flush=True)
print('done', n, 'iterations,', bad, 'mismatching solves')
