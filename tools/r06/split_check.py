#!/usr/bin/env python3
"""Is the final S of a C5 volume independent of how the batch is split (2 + 2 against 4), with and without the remainder cut?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from xinvert_amd import synthetic
from xinvert_amd.resident import ResidentProblem
import torch
p = synthetic.omega_latlon(50, 360, 720, steps=4)
for sweeps in (10, 11, 4):
    for cus in (0, -1):
        res = {}
        for name, blocks in (('4', [(0, 4)]), ('2+2', [(0, 2), (2, 4)]), ('1x4', [(0, 1), (1, 2), (2, 3), (3, 4)])):
            out, sts = [], []
            for lo, hi in blocks:
                rp = ResidentProblem(p, members=(lo, hi))
                fl, st = rp.solve(sweeps - 1, 0.0, cu_count=cus)
                out.append(rp.result()); sts.append((st['k_chunks'], st['cut_tiles'], st['lanes']))
                del rp
            res[name] = (np.concatenate(out), sts)
        a = res['4'][0]
        for name in ('2+2', '1x4'):
            b = res[name][0]
            print('sweeps', sweeps, 'cus', cus, name, 'vs 4:', 'equal' if np.array_equal(a, b, equal_nan=True) else 'DIFFER at %d points' % int((a != b).sum()),
                  res['4'][1], res[name][1], flush=True)
