#!/usr/bin/env python3
"""Host-pointer batched entry against the device-resident entry on the same batch: how much of the
PCIe traffic the chunked upload / solve / download pipeline hides.
  python tools/bench_host_pipeline.py [c5|c4] [--members M] [--sweeps S]"""
import argparse, ctypes, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from xinvert_amd import _lib, synthetic
from xinvert_amd.resident import ResidentProblem, FN, scalars

ap = argparse.ArgumentParser()
ap.add_argument('config', nargs='?', default='c5')
ap.add_argument('--members', type=int, default=15)
ap.add_argument('--sweeps', type=int, default=200)
ap.add_argument('--chunks', default='', help='comma list of host_chunk values to time (default: a sweep over several)')
ap.add_argument('--reps', type=int, default=3)
ap.add_argument('--lanes', type=int, default=0)
ap.add_argument('--inflight', default='0', help='comma list of xinv_options.host_inflight values')
ap.add_argument('--check-every', type=int, default=0, help='xinv_options.check_every: launches per host poll (0 = the engine chooses)')
a = ap.parse_args()
p = (synthetic.omega_latlon(50, 360, 720, a.members) if a.config == 'c5' else synthetic.poisson_latlon(1800, 3600, members=a.members) if a.config == 'c2'
     else synthetic.gill_matsuno(720, 1440, a.members))
L = _lib.require_gpu()
nb = p['S0'].shape[0]; n = int(np.prod(p['S0'].shape[1:]))
rp = ResidentProblem(p)
best = 1e9
for _ in range(3):
    rp.reset(); t = time.perf_counter(); rp.solve(a.sweeps - 1, 0.0); best = min(best, time.perf_counter() - t)
dev_s = best
print(json.dumps({'entry': 'dev (inputs resident)', 'solve_ms': dev_s * 1e3, 'point_sweeps_per_s': nb * n * a.sweeps / dev_s}), flush=True)
del rp

shared = set(p['shared'])
arrs = [np.ascontiguousarray(p['S0'], dtype=np.float64)]
strides = [n]
for k, c in enumerate(p['coefs']):
    c = np.ascontiguousarray(c, dtype=np.float64)
    null = (k == 1 and p['kind'] in ('std2d', 'gen2d') and not c.any())
    arrs.append(None if null else c); strides.append(0 if k in shared else n)
for chunk, infl in [(c_, i_) for c_ in ([int(v) for v in a.chunks.split(',')] if a.chunks else (nb, 0, 1, 2, 3, 5))
                    for i_ in [int(v) for v in a.inflight.split(',')]]:
    best = 1e9
    for rep in range(a.reps):
        S = arrs[0].copy()
        fl = np.tile(np.array([0., 1., 0.]), (nb, 1))
        o = _lib.options(host_chunk=chunk, host_inflight=infl, lanes=a.lanes, check_every=a.check_every)
        t = time.perf_counter()
        rc = getattr(L, FN[p['kind']] + '_batched')(_lib.hptr(S), *[_lib.hptr(x) for x in arrs[1:]], nb, _lib.strides_arg(strides),
                                                    *scalars(p), _lib.hptr(fl), a.sweeps - 1, 0.0, ctypes.byref(o))
        dt = time.perf_counter() - t
        _lib.check(rc)
        if dt < best:
            best, st = dt, _lib.last_stats()
    print(json.dumps({'entry': 'host pointers', 'host_chunk': chunk, 'inflight': infl, 'check_every': a.check_every, 'lanes': a.lanes, 'chunks': st['host_chunks'], 'wall_ms': best * 1e3,
                      'h2d_ms': st['h2d_ms'], 'd2h_ms': st['d2h_ms'], 'vs_dev': best / dev_s,
                      'point_sweeps_per_s': nb * n * a.sweeps / best}), flush=True)
