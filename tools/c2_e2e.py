#!/usr/bin/env python3
"""apps.invert_Poisson host to host on the headline slice (3600 x 1800, 500 sweeps): wall / library / copy spans, best and median.
  python tools/c2_e2e.py REPS"""
import os
import sys, time, json, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import xinvert_amd as xa
from xinvert_amd import synthetic
p = synthetic.poisson_latlon(1800, 3600)
F = xa.Field(p['zeta'][0], ('lat', 'lon'), {'lat': p['lat'], 'lon': p['lon']})
iP = {'BCs': [p['BCy'], p['BCx']], 'mxLoop': 499, 'tolerance': 0.0, 'printInfo': False, 'device_prep': True}
xa.invert_Poisson(F, ['lat', 'lon'], iParams=iP)
ts = []
for _ in range(int(sys.argv[1])):
    t = time.perf_counter(); S = xa.invert_Poisson(F, ['lat', 'lon'], iParams=iP); dt = time.perf_counter() - t
    st = S.iParams['stats']; ts.append((dt * 1e3, st['wall_ms'], st['h2d_ms'], st['d2h_ms'], st['plan_ms'], st['sweep_ms']))
ts.sort()
print(json.dumps({'C2_invert_Poisson wall/library/h2d/d2h/plan/sweep ms, best and median': [ [round(v, 3) for v in ts[0]], [round(v, 3) for v in ts[len(ts) // 2]] ]}))
