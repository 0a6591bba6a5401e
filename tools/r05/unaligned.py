#!/usr/bin/env python3
"""What an ODD row length costs the streaming kernels without any seam (fixed x): strips are not 16-byte aligned and a lane's
column pair is read as two 8-byte loads.  Nine-point forms (k_fused9) and the 3-D standard form (k_fused3d / k_pipe3d)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import util                                                   # noqa: E402

for kind in ('std2d', 'gen2d'):
    for yc, xc in ((2000, 2000), (2000, 2001)):
        p = util.rand2d(kind, yc, xc, 'fixed', 'fixed', bnz=True, seed=1)
        best = 0.0
        for rep in range(3):
            S, fl, st = util.run_hip_dev([p], 199, 0.0, timing=1)
            best = max(best, yc * xc * 200 / (st['sweep_ms'] * 1e-3))
        print('%s nine-point fixed x %dx%d %.3g point-sweeps/s (path %d, %d sweeps per pass)' % (kind, yc, xc, best, st['path'], st['sweeps_per_launch']))
for zc, yc, xc in ((50, 360, 720), (50, 360, 721)):
    for bcx in ('fixed', 'periodic'):
        p = util.rand3d(zc, yc, xc, 'fixed', bcx, seed=1)
        for k in range(3):                                    # coefficients constant along x, as every lat-lon omega problem
            p['coefs'][k] = np.ascontiguousarray(np.broadcast_to(p['coefs'][k][:, :, :1], (zc, yc, xc)))
        for spl in (0, 1):
            best = 0.0
            for rep in range(3):
                S, fl, st = util.run_hip_dev([p] * 4, 49, 0.0, timing=1, sweeps_per_launch=spl)
                best = max(best, 4 * zc * yc * xc * 50 / (st['sweep_ms'] * 1e-3))
            print('std3d %s x %dx%dx%d x 4 spl=%d: %.3g point-sweeps/s (path %d, %d sweeps per pass)' % (bcx, zc, yc, xc, spl, best, st['path'], st['sweeps_per_launch']))
