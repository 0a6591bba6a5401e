"""Stress of the wave-pipelined pass: alternate solves whose per-row factors differ and count mismatches
against the oracle (python tools/stress_pipe.py [iterations])."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np
from util import rand2d, run_oracle, run_hip_batched
from test_gpu_parity import _uniform2d

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
cases = []
for seed, shape, bc in ((1, (33, 257), ('fixed', 'fixed')), (2, (33, 257), ('fixed', 'fixed')),
                        (3, (40, 300), ('fixed', 'periodic')), (4, (33, 257), ('extend', 'fixed')),
                        (5, (64, 512), ('fixed', 'periodic'))):
    p = _uniform2d(rand2d('std2d', shape[0], shape[1], bc[0], bc[1], 0, seed & 1, seed=seed))
    So, flo = run_oracle(p, 24, 1e-9, 2)
    cases.append((p, So, flo))
bad = 0
for it in range(n):
    for ci, (p, So, flo) in enumerate(cases):
        for rows in (16, 0):
            S, fl, st = run_hip_batched([p], 24, 1e-9, path=2, sweeps_per_launch=4, rows_per_tile=rows)
            if not np.array_equal(S[0], So):
                bad += 1
                d = (S[0] != So)
                rowsbad = np.where(d.any(axis=1))[0]
                print('iter', it, 'case', ci, 'rows', rows, 'mismatch points', int(d.sum()), 'rows', rowsbad[:6], '...', rowsbad[-3:],
                      'cols', np.where(d.any(axis=0))[0][[0, -1]], flush=True)
print('done', n, 'iterations,', bad, 'mismatching solves')
