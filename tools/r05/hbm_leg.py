#!/usr/bin/env python3
"""The HBM-bound variant of bench.py (k_fused2d K = 1, 8 members of 3600x1800 each with its own A, C, F: 2.07 GB, every tile):
pass time and streamed TB/s.   XINV_SO selects the library."""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from xinvert_amd import synthetic
from xinvert_amd.resident import ResidentProblem
import torch
hm = int(sys.argv[1]) if len(sys.argv) > 1 else 8
p = synthetic.poisson_latlon(1800, 3600, mask=True, members=hm)
p = dict(p); p['coefs'] = [np.broadcast_to(c, (hm,) + c.shape) if k in (0, 2) else c for k, c in enumerate(p['coefs'])]; p['shared'] = (1,)
rp = ResidentProblem(p)
o = dict(sweeps_per_launch=1, timing=1, no_xuniform=1, no_tile_skip=1)
for lanes in (0, 1):
    rp.reset(); rp.solve(29, 0.0, lanes=lanes, **o)
    ms = nl = 0
    for _ in range(3):
        rp.reset(); fl, s = rp.solve(59, 0.0, lanes=lanes, **o); ms += s['sweep_ms']; nl += s['sweep_launches']
    avg = ms / nl
    print(json.dumps({'so': os.path.basename(os.environ.get('XINV_SO', 'shipped')), 'members': hm, 'lanes': s['lanes'], 'pass_us': avg * 1e3,
                      'streamed_TBps': 40.0 * hm * 6.48e6 / (avg * 1e-3) / 1e12, 'frac_of_8TBps': 40.0 * hm * 6.48e6 / (avg * 1e-3) / 8e12}))
