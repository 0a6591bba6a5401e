#!/usr/bin/env python3
"""A/B timing of the 3-D standard form (BASELINE configs[4] shape): one sweep per pass against two.
  python tools/ab_3d.py [members ...]      XINV_SO selects the build, XINV_3D_K2=0 the one-sweep kernel"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from xinvert_amd import synthetic
from xinvert_amd.resident import ResidentProblem
import torch
def run(name, p, sweeps, **o):
    rp = ResidentProblem(p)
    best = None
    for rep in range(3):
        rp.reset(); torch.cuda.synchronize(); t = time.perf_counter()
        fl, s = rp.solve(sweeps - 1, 0.0, timing=1, **o)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
        best = dt if best is None or dt < best else best
    print('%-14s %-20s %-26s value %.4g  launch %.1f us  K %d' % (os.path.basename(os.environ.get('XINV_SO', '') or 'main'), name, o, rp.nb * rp.n * sweeps / best, s['sweep_ms'] / s['sweep_launches'] * 1e3, s['sweeps_per_launch']), flush=True)
for mem in [int(x) for x in sys.argv[1:] if x.isdigit()] or [15]:
    p = synthetic.omega_latlon(50, 360, 720, mem)
    run('C5 %d volumes' % mem, p, 50)
    run('C5 %d volumes' % mem, p, 50, sweeps_per_launch=2)
if 'ofes' in sys.argv:
    p = synthetic.omega_latlon(601, 300, 300, 1)
    run('ofes 601x300x300', p, 100)
    run('ofes 601x300x300', p, 100, sweeps_per_launch=2)
