#!/usr/bin/env python3
"""3-D standard form, x-uniform coefficients, BCy = 'extend' beside 'fixed': what the engine chooses (two sweeps per pass where k_pipe3d takes the problem) and the one-sweep
kernel (k_fused3d), 8 volumes of 50 x 360 x 720, 50 sweeps (HIP-event time of the sweep launches)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import util                                                   # noqa: E402
zc, yc, xc = 50, 360, 720
for bcx in ('periodic', 'fixed'):
    for bcy in ('fixed', 'extend'):
        p = util.rand3d(zc, yc, xc, bcy, bcx, seed=1)
        for k in range(3):
            p['coefs'][k] = np.ascontiguousarray(np.broadcast_to(p['coefs'][k][:, :, :1], (zc, yc, xc)))
        for spl in (0, 1):
            best = 0
            for rep in range(3):
                S, fl, st = util.run_hip_dev([p] * 8, 49, 0.0, timing=1, sweeps_per_launch=spl)
                best = max(best, 8 * zc * yc * xc * 50 / (st['sweep_ms'] * 1e-3))
            print('BCx %s BCy %s: %.3g point-sweeps/s, %d sweeps per pass' % (bcx, bcy, best, st['sweeps_per_launch']))
