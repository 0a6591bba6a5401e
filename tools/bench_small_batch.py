#!/usr/bin/env python3
"""Many small slices in one call (the reference's typical use: a year of daily 2.5-degree fields).
  python tools/bench_small_batch.py [--path 0|1|2]      0 = engine's choice, 1 = colour passes, 2 = streaming kernels"""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from xinvert_amd import synthetic
from xinvert_amd.resident import ResidentProblem
ap = argparse.ArgumentParser(); ap.add_argument('--path', type=int, default=0); ap.add_argument('--sweeps', type=int, default=400)
a = ap.parse_args()
CASES = [('gm', 73, 144, 1), ('gm', 73, 144, 365), ('gm', 73, 144, 3650), ('poisson', 72, 144, 365),
         ('poisson', 180, 360, 1), ('poisson', 180, 360, 64), ('gm', 180, 360, 365)]
for (kind, ny, nx, nb) in CASES:
    p = synthetic.gill_matsuno(ny, nx, nb) if kind == 'gm' else synthetic.poisson_latlon(ny, nx, mask=False, BCs=('extend', 'periodic'), members=nb)
    rp = ResidentProblem(p)
    best = 1e9
    for rep in range(3):
        rp.reset()
        t = time.perf_counter(); fl, st = rp.solve(a.sweeps - 1, 0.0, timing=1, path=a.path); best = min(best, time.perf_counter() - t)
    print(json.dumps({'case': kind, 'shape': [nb, ny, nx], 'point_sweeps_per_s': nb * ny * nx * a.sweeps / best, 'solve_ms': best * 1e3,
                      'us_per_sweep': best * 1e6 / a.sweeps, 'path': st['path'], 'rows_per_tile': st['rows_per_tile'], 'um': st['xuniform_mask']}), flush=True)
    del rp
