#!/usr/bin/env python3
"""Throughput of every BASELINE.json configuration on one GPU (not the bench.py contract line), through the same path as
bench.py's per-configuration lines: a ResidentProblem solving on a resident plan.

  python tools/bench_configs.py [c1 c2 c3 c3m c4 c5 ...] [--sweeps N] [--spl K] [--rows R] [--members M] [--lanes L]
                                [--no-plan] [--events]
Prints one JSON object per configuration: point-sweeps/s over the whole solve (inputs resident in HBM), mean pass time
from HIP events (timing = 1: events around chunks of launches), algorithmic GB/s.  --events (one lane only): a second
run with timing = 2 -- an event behind EVERY sweep launch on the launches' own stream -- gives per-launch min / avg / max
without a profiler's per-dispatch overhead (profiles/r05_launch_events_*.txt; the events themselves cost ~1 us per launch,
so `value` is taken from the timing = 1 run)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

ALG = {'std2d': 48, 'gen2d': 72, 'std3d': 48, 'bih2d': 96, 'gen3d': 80}


def make(name, a):
    from xinvert_amd import synthetic
    if name == 'c1':
        return synthetic.poisson_latlon(180, 360, mask=False, members=a.members or 1), a.sweeps or 500
    if name == 'c2':
        return synthetic.poisson_latlon(1800, 3600, mask=True, members=a.members or 1), a.sweeps or 200
    if name == 'c3':
        return synthetic.stommel_cartesian(2000, 2000), a.sweeps or 200
    if name == 'c3m':
        return synthetic.munk_cartesian(2000, 2000), a.sweeps or 100
    if name == 'c3mxy':                                  # Munk with A4(x, y), R(x, y): the vector-stream variant of k_fusedbih
        return synthetic.munk_cartesian(2000, 2000, varying=True), a.sweeps or 100
    if name == 'c4':
        return synthetic.gill_matsuno(720, 1440, a.members or 8), a.sweeps or 200
    if name == 'gm73':                                  # the reference's own regime: tests/test_GillMatsuno.py:14-57
        return synthetic.gill_matsuno(73, 144, a.members or 1), a.sweeps or 600
    if name == 'c5':
        return synthetic.omega_latlon(50, 360, 720, a.members or 2), a.sweeps or 50
    if name == 'c5odd':                                  # odd width, periodic x: the seam inside k_fused3d
        return synthetic.omega_latlon(50, 360, 721, a.members or 2), a.sweeps or 50
    if name == 'c5g':
        return synthetic.ocean3d_latlon(50, 360, 720, a.members or 2), a.sweeps or 50
    if name.startswith('poisson:') or name.startswith('gm:'):       # poisson:1800x3601 / gm:720x1441 -- any slice shape (seam rates)
        ny, nx = (int(v) for v in name.split(':')[1].split('x'))
        if name.startswith('gm:'):
            return synthetic.gill_matsuno(ny, nx, a.members or 1), a.sweeps or 400
        return synthetic.poisson_latlon(ny, nx, mask=a.mask, members=a.members or 1, BCs=('fixed', a.bcx)), a.sweeps or 400
    if name == 'ofes':
        # the shape of the reference's only published wall-clock figure: invert_omega on a
        # 601 x 300 x 300 ocean grid, 501 sweeps, 730 s per solve on one CPU core
        # (docs/source/notebooks/11_Omega_equation.ipynb:525-529, tests/test_OmegaEq.py:188-222)
        return synthetic.omega_latlon(601, 300, 300, a.members or 1), a.sweeps or 501
    raise SystemExit('unknown config ' + name)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('configs', nargs='*', default=['c1', 'c2', 'c3', 'c4', 'c5'])
    ap.add_argument('--sweeps', type=int, default=0)
    ap.add_argument('--spl', type=int, default=0)
    ap.add_argument('--rows', type=int, default=0)
    ap.add_argument('--path', type=int, default=0)
    ap.add_argument('--members', type=int, default=0)
    ap.add_argument('--lanes', type=int, default=0)
    ap.add_argument('--bcy', default='', help='override BCy of the configuration (c5 --bcy extend)')
    ap.add_argument('--cus', type=int, default=0, help='xinv_options.cu_count (0 = the device; a huge count = no remainder cut in k_pipe3d)')
    ap.add_argument('--reps', type=int, default=3)
    ap.add_argument('--bcx', default='periodic', help='poisson:<ny>x<nx>: BCx')
    ap.add_argument('--mask', action='store_true', help='poisson:<ny>x<nx>: with the synthetic land/sea mask')
    ap.add_argument('--no-xuniform', action='store_true', help='stream every coefficient array in full')
    ap.add_argument('--no-pipe', action='store_true')
    ap.add_argument('--no-plan', action='store_true', help='every solve through xinv_<form>_f64_dev (re-derives everything)')
    ap.add_argument('--events', action='store_true', help='one more run with an event behind every launch (timing = 2)')
    a = ap.parse_args()
    import torch
    from xinvert_amd import _lib
    from xinvert_amd.resident import ResidentProblem
    _lib.require_gpu()
    for name in a.configs:
        p, sw = make(name, a)
        if a.bcy:
            p['BCy'] = a.bcy
        rp = ResidentProblem(p, plan=not a.no_plan)
        nb, n = rp.nb, rp.n
        opt = dict(sweeps_per_launch=a.spl, rows_per_tile=a.rows, path=a.path, no_xuniform=1 if a.no_xuniform else 0,
                   no_pipe=1 if a.no_pipe else 0, lanes=a.lanes, cu_count=a.cus)
        best = None
        for rep in range(a.reps + 1):
            rp.reset(); torch.cuda.synchronize()
            t = time.perf_counter()
            fl, st = rp.solve(sw - 1, 0.0, timing=1, **opt)
            dt = time.perf_counter() - t
            if rep and (best is None or dt < best[0]):
                best = (dt, st)
        dt, st = best
        k = st['sweeps_per_launch']
        avg_ms = st['sweep_ms'] / max(st['sweep_launches'], 1)
        out = {'config': name, 'kind': p['kind'], 'shape': [nb] + list(rp.core), 'sweeps': sw,
               'point_sweeps_per_s': nb * n * sw / dt, 'solve_ms': dt * 1e3, 'planned': st['planned'],
               'path': st['path'], 'colours': st['colours'], 'xuniform_mask': st['xuniform_mask'], 'sweeps_per_launch': k,
               'rows_per_tile': st['rows_per_tile'], 'k_chunks': st.get('k_chunks'), 'cut_tiles': st.get('cut_tiles'), 'masked_tile_pct': st['masked_tile_pct'], 'lanes': st['lanes'],
               'pipelined': st['pipelined'], 'launches': st['sweep_launches'], 'avg_launch_ms': avg_ms,
               'launches_x_avg_ms': avg_ms * st['sweep_launches'],
               'alg_GBps': ALG[p['kind']] * nb * n * k / (avg_ms * 1e-3) / 1e9 if st['path'] == 2
               else ALG[p['kind']] * nb * n * sw / (st['sweep_ms'] * 1e-3) / 1e9,
               'flags0': fl[0].tolist()}
        if a.events:
            rp.reset(); torch.cuda.synchronize()
            fl, s2 = rp.solve(sw - 1, 0.0, timing=2, **opt)
            out['launch_events'] = {'lanes': s2['lanes'], 'min_us': s2['launch_us_min'], 'avg_us': s2['launch_us_avg'],
                                    'max_us': s2['launch_us_max'],
                                    'note': 'timing = 2: an event behind every launch (one lane; 0 with several: the chains overlap)'}
        print(json.dumps(out), flush=True)
        rp.close()
        del rp
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
