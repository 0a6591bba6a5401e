"""Timing-only experiments on the headline workload (results discarded): run with XINV_EXP_NOCTL=1 (no norm,
no stop rule, every launch runs) and any library variant (XINV_SO=...).  Prints the mean sweep-launch time.
    XINV_EXP_NOCTL=1 XINV_SO=build/libxinv_nobar.so python tools/time_noctl.py [--rows N] [--sweeps S]"""
import argparse, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
ap = argparse.ArgumentParser()
ap.add_argument('--rows', type=int, default=0)
ap.add_argument('--spl', type=int, default=0)
ap.add_argument('--sweeps', type=int, default=500)
ap.add_argument('--reps', type=int, default=5)
a = ap.parse_args()
import torch
from xinvert_amd import _lib, synthetic
from xinvert_amd.resident import ResidentProblem
p = synthetic.poisson_latlon(1800, 3600, mask=True, seed=0)
rp = ResidentProblem(p, device=0)
best = None
for rep in range(a.reps):
    rp.reset()
    try:
        fl, s = rp.solve(a.sweeps - 1, 0.0, sweeps_per_launch=a.spl, rows_per_tile=a.rows, timing=1)
    except _lib.XinvError:
        s = _lib.last_stats()
    us = 1e3 * s['sweep_ms'] / max(1, s['sweep_launches'])
    best = us if best is None else min(best, us)
print('launch %.1f us (best of %d), K=%d RY=%d launches=%d' % (best, a.reps, s['sweeps_per_launch'], s['rows_per_tile'], s['sweep_launches']))
