#!/usr/bin/env python3
"""Throughput of the 9-point (B != 0) forms, which run on the 4-colour path."""
import ctypes, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from xinvert_amd import _lib
import util
L = _lib.require_gpu()
dev = torch.device('cuda', 0)
ROWS = int(os.environ.get('NINE_ROWS', '0'))
for kind, spl, path in (('std2d', 1, 0), ('std2d', 2, 0), ('std2d', 3, 0), ('std2d', 0, 0), ('std2d', 0, 1), ('gen2d', 1, 0), ('gen2d', 2, 0), ('gen2d', 0, 0), ('gen2d', 0, 1)):
    ny = nx = 2000
    p = util.rand2d(kind, ny, nx, 'fixed', 'periodic', bnz=True, msk=False, seed=1, omega=0.9)
    p['coefs'][1] = p['coefs'][1] * 0.2            # weak cross term: keeps the iteration stable
    n = ny * nx
    S0 = torch.from_numpy(p['S0'][None].copy()).to(dev); S = S0.clone()
    cs = [torch.from_numpy(np.ascontiguousarray(c)).to(dev) for c in p['coefs']]
    strides = [n] * (1 + len(cs))
    fl = np.tile(np.array([0., 1., 0.]), (1, 1)); sw = 100
    opt = _lib.options(timing=1, sweeps_per_launch=spl, path=path, rows_per_tile=ROWS)
    args = [ctypes.c_void_p(S.data_ptr())] + [ctypes.c_void_p(c.data_ptr()) for c in cs] + \
           [1, _lib.strides_arg(strides)] + util._scal(p, fl, sw - 1, 0.0) + [ctypes.byref(opt), None]
    fn = getattr(L, util._FN[kind] + '_dev')
    best = 1e9
    for rep in range(3):
        S.copy_(S0); torch.cuda.synchronize()
        t = time.perf_counter(); _lib.check(fn(*args)); best = min(best, time.perf_counter() - t)
    st = _lib.last_stats()
    print(json.dumps({'kind': kind, 'spl': st['sweeps_per_launch'], 'flags': fl[0].tolist(), 'shape': [ny, nx], 'point_sweeps_per_s': n * sw / best, 'path': st['path'],
                      'colours': st['colours'], 'rows': st['rows_per_tile'], 'sweep_ms': st['sweep_ms'] / sw}))
