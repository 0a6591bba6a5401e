#!/usr/bin/env python3
"""FusedGen2DQ (general form, A C G varying along x, point-factor stream) on the pipelined pass: parity against the oracle
and rate against k_fused2d at three sweeps per pass.   python tools/r05/pq_check.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import util
from oracle import COLOUR_2


def uni_def(p):
    q = dict(p); cs = [np.array(c, copy=True) for c in p['coefs']]
    for k in (3, 4, 5):                                   # D, E, F constant along x
        cs[k] = np.repeat(cs[k][:, :1], cs[k].shape[1], axis=1)
    q['coefs'] = cs
    return q


bad = 0
for (yc, xc, BCy, BCx, msk) in [(90, 420, 'fixed', 'periodic', 1), (200, 700, 'extend', 'fixed', 1), (333, 1000, 'fixed', 'fixed', 0),
                                (64, 130, 'extend', 'periodic', 1)]:
    ps = [uni_def(util.rand2d('gen2d', yc, xc, BCy, BCx, 0, msk, seed=5 + m)) for m in range(2)]
    for mx, tol in ((21, 0.0), (300, 1e-3)):
        S, fl, st = util.run_hip_dev(ps, mx, tol)
        S1, fl1, st1 = util.run_hip_dev(ps, mx, tol, no_point_factor=1)
        assert np.array_equal(S, S1) and np.array_equal(fl, fl1), 'Q stream differs from the in-kernel factor'
        for m, q in enumerate(ps):
            So, flo = util.run_oracle(q, mx, tol, COLOUR_2)
            ok = np.array_equal(S[m], So) and fl[m][2] == flo[2]
            bad += (not ok)
            print(yc, xc, BCy, BCx, msk, mx, tol, 'member', m, 'pipelined', st['pipelined'], 'um', st['xuniform_mask'], 'K', st['sweeps_per_launch'], 'OK' if ok else 'MISMATCH %d' % int((S[m] != So).sum()), int(flo[2]))
print('failures:', bad)

from xinvert_amd import synthetic
from xinvert_amd.resident import ResidentProblem
import torch
p = synthetic.stommel_cartesian(2000, 2000)
for nq, spl in ((1, 0), (0, 0), (1, 0), (0, 0), (1, 2), (0, 2)):
    rp = ResidentProblem(p)
    for _ in range(2):
        rp.reset(); rp.solve(499, 0.0, sweeps_per_launch=spl, timing=1, no_point_factor=nq)
    rp.reset(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5):
        fl, s = rp.solve(499, 0.0, sweeps_per_launch=spl, timing=1, no_point_factor=nq)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print('C3 Stommel point-factor stream %s spl=%d: %.4g point-sweeps/s, launch %.2f us, K=%d rows=%d' % ('off' if nq else 'ON', spl, 4e6 * 500 * 5 / dt, s['sweep_ms'] / s['sweep_launches'] * 1e3, s['sweeps_per_launch'], s['rows_per_tile']))
    rp.close()
