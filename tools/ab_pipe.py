#!/usr/bin/env python3
"""A/B timing of the pipelined pass on the headline workload: XINV_SO selects the library build.
  python tools/ab_pipe.py [rows_per_tile ...]"""
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
    print('%-18s %-20s %-22s value %.4g  launch %.1f us  K %d rows %d pipe %d' % (os.path.basename(os.environ.get('XINV_SO', '') or 'main'), name, o, rp.nb * rp.n * sweeps / best, s['sweep_ms'] / s['sweep_launches'] * 1e3, s['sweeps_per_launch'], s['rows_per_tile'], s.get('pipelined', -1)), flush=True)
c2 = synthetic.poisson_latlon(1800, 3600, mask=True)
rows = [int(x) for x in sys.argv[1:] if x.isdigit()] or [0]
for r in rows:
    run('C2 3600x1800', c2, 500, **({'rows_per_tile': r} if r else {}))
if 'c4' in sys.argv:
    run('C4 64', synthetic.gill_matsuno(720, 1440, 64), 200)
if 'c1' in sys.argv:
    run('C1 360x180', synthetic.poisson_latlon(180, 360, mask=False), 500)
