#!/usr/bin/env python3
"""Odd widths (periodic x) beside even widths for the 9-point forms (k_fused9's SEAM variants) and the colour launches:
point-sweeps/s from the HIP-event time of the sweep launches (profiles/r04_seam_rates.txt, last block)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import util                                                   # noqa: E402

for kind in ('std2d', 'gen2d'):
    for yc, xc in ((2000, 2000), (2000, 2001), (720, 1441)):
        p = util.rand2d(kind, yc, xc, 'fixed', 'periodic', bnz=True, seed=1)
        for name, opt in (('streaming', dict()), ('colour launches', dict(path=1))):
            best = 0.0
            for rep in range(3):
                S, fl, st = util.run_hip_dev([p], 199, 0.0, timing=1, **opt)
                best = max(best, yc * xc * 200 / (st['sweep_ms'] * 1e-3))
            print('%s nine-point %dx%d %-16s %.3g point-sweeps/s (path %d, %d colours, %d sweeps per pass)'
                  % (kind, yc, xc, name, best, st['path'], st['colours'], st['sweeps_per_launch']))
