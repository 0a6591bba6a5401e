#!/usr/bin/env python3
"""Throughput of every BASELINE.json configuration on one GPU (not the bench.py contract line).

  python tools/bench_configs.py [c1 c2 c3 c4 c5] [--sweeps N] [--spl K] [--rows R] [--members M]
Prints one JSON object per configuration: point-sweeps/s over the whole solve (inputs resident
in HBM), mean sweep-launch time from HIP events, algorithmic GB/s.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

ALG = {'std2d': 48, 'gen2d': 72, 'std3d': 48, 'bih2d': 96, 'gen3d': 80}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('configs', nargs='*', default=['c1', 'c2', 'c3', 'c4', 'c5'])
    ap.add_argument('--sweeps', type=int, default=0)
    ap.add_argument('--spl', type=int, default=0)
    ap.add_argument('--rows', type=int, default=0)
    ap.add_argument('--path', type=int, default=0)
    ap.add_argument('--members', type=int, default=0)
    ap.add_argument('--reps', type=int, default=3)
    ap.add_argument('--no-xuniform', action='store_true', help='stream every coefficient array in full')
    a = ap.parse_args()
    import torch
    from xinvert_amd import _lib, synthetic
    import util
    L = _lib.require_gpu()
    dev = torch.device('cuda', 0)
    for name in a.configs:
        if name == 'c1':
            p = synthetic.poisson_latlon(180, 360, mask=False, members=a.members or 1); sw = a.sweeps or 500
        elif name == 'c2':
            p = synthetic.poisson_latlon(1800, 3600, mask=True, members=a.members or 1); sw = a.sweeps or 200
        elif name == 'c3':
            p = synthetic.stommel_cartesian(2000, 2000); sw = a.sweeps or 200
        elif name == 'c3m':
            p = synthetic.munk_cartesian(2000, 2000); sw = a.sweeps or 100
        elif name == 'c4':
            p = synthetic.gill_matsuno(720, 1440, a.members or 8); sw = a.sweeps or 200
        elif name == 'c5':
            p = synthetic.omega_latlon(50, 360, 720, a.members or 2); sw = a.sweeps or 50
        elif name == 'c5odd':                              # odd width, periodic x: the seam inside k_fused3d
            p = synthetic.omega_latlon(50, 360, 721, a.members or 2); sw = a.sweeps or 50
        elif name == 'c5g':
            p = synthetic.ocean3d_latlon(50, 360, 720, a.members or 2); sw = a.sweeps or 50
        elif name == 'ofes':
            # the shape of the reference's only published wall-clock figure: invert_omega on a
            # 601 x 300 x 300 ocean grid, 501 sweeps, 730 s per solve on one CPU core
            # (docs/source/notebooks/11_Omega_equation.ipynb:525-529, tests/test_OmegaEq.py:188-222)
            p = synthetic.omega_latlon(601, 300, 300, a.members or 1); sw = a.sweeps or 501
        else:
            raise SystemExit('unknown config ' + name)
        nb = p['S0'].shape[0]
        n = int(np.prod(p['S0'].shape[1:]))
        S0 = torch.from_numpy(np.ascontiguousarray(p['S0'], dtype=np.float64)).to(dev)
        S = S0.clone()
        cs = [torch.from_numpy(np.ascontiguousarray(c, dtype=np.float64)).to(dev) for c in p['coefs']]
        strides = [n] + [0 if k in p['shared'] else n for k in range(len(cs))]
        fl = np.tile(np.array([0., 1., 0.]), (nb, 1))
        opt = _lib.options(sweeps_per_launch=a.spl, rows_per_tile=a.rows, path=a.path, timing=1,
                           no_xuniform=1 if a.no_xuniform else 0)
        fn = getattr(L, util._FN[p['kind']] + '_dev')
        args = [ctypes.c_void_p(S.data_ptr())] + [ctypes.c_void_p(c.data_ptr()) for c in cs] + \
               [nb, _lib.strides_arg(strides)] + util._scal(p, fl, sw - 1, 0.0) + [ctypes.byref(opt), None]
        best = None
        for rep in range(a.reps + 1):
            S.copy_(S0); torch.cuda.synchronize()
            t = time.perf_counter()
            _lib.check(fn(*args))
            dt = time.perf_counter() - t
            st = _lib.last_stats()
            if rep and (best is None or dt < best[0]):
                best = (dt, st)
        dt, st = best
        k = st['sweeps_per_launch']
        avg_ms = st['sweep_ms'] / max(st['sweep_launches'], 1)
        print(json.dumps({'config': name, 'kind': p['kind'], 'shape': list(p['S0'].shape), 'sweeps': sw,
                          'point_sweeps_per_s': nb * n * sw / dt, 'solve_ms': dt * 1e3,
                          'path': st['path'], 'colours': st['colours'], 'xuniform_mask': st['xuniform_mask'], 'sweeps_per_launch': k, 'rows_per_tile': st['rows_per_tile'], 'masked_tile_pct': st['masked_tile_pct'],
                          'avg_launch_ms': avg_ms,
                          'alg_GBps': ALG[p['kind']] * nb * n * k / (avg_ms * 1e-3) / 1e9 if st['path'] == 2
                          else ALG[p['kind']] * nb * n * sw / (st['sweep_ms'] * 1e-3) / 1e9,
                          'flags0': fl[0].tolist()}))
        del S, S0, cs
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
