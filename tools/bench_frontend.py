#!/usr/bin/env python3
"""End-to-end time of the reference-shaped front end (invert_Poisson on host arrays) at
BASELINE configs[1]: where the wall-clock goes outside the sweep kernels."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import xinvert_amd as xa
    from xinvert_amd import _lib
    ny, nx = 1800, 3600
    lat = np.linspace(-89.95, 89.95, ny); lon = np.arange(nx) * 0.1
    rng = np.random.default_rng(1)
    la, lo = np.deg2rad(lat)[:, None], np.deg2rad(lon)[None, :]
    vor = 1e-5 * (np.sin(3 * lo) * np.cos(2 * la) + 0.3 * np.cos(7 * lo + 1) * np.sin(5 * la))
    vor[(np.sin(4 * lo + 2 * la) > 0.4) & (np.abs(la) < 1.2)] = np.nan
    F = xa.Field(vor, ('lat', 'lon'), {'lat': lat, 'lon': lon})
    iP = {'BCs': ['fixed', 'periodic'], 'mxLoop': int(sys.argv[1]) if len(sys.argv) > 1 else 499,
          'tolerance': 0.0, 'printInfo': False}
    for prep in (True, False):
        iP['device_prep'] = prep
        xa.invert_Poisson(F, ['lat', 'lon'], iParams=iP)          # warm-up (library load, pools)
        best = 1e9
        for _ in range(5):
            t = time.perf_counter()
            S = xa.invert_Poisson(F, ['lat', 'lon'], iParams=iP)
            best = min(best, time.perf_counter() - t)
        st = S.iParams['stats']
        print('invert_Poisson %dx%d, %d sweeps, mask/scale/de-mask on the %s: %.1f ms end to end (library call %.1f ms: h2d %.1f, sweeps %.1f, d2h %.1f), masked tiles %d%%'
              % (nx, ny, iP['mxLoop'] + 1, 'device' if prep else 'host (numpy)', best * 1e3, st['wall_ms'], st['h2d_ms'], st['sweep_ms'], st['d2h_ms'], st['masked_tile_pct']))
    # the same forcing as float32 (the dtype of every dataset the reference ships): it travels as float32 and is promoted
    # on the device (xinv_options.f32_mask); with float32_out the solution comes back as float32 too
    iP['device_prep'] = True
    F32 = xa.Field(vor.astype(np.float32), ('lat', 'lon'), {'lat': lat, 'lon': lon})
    for tag, extra in (('float32 forcing', {}), ('float32 forcing, float32 solution', {'float32_out': True})):
        q = dict(iP); q.update(extra)
        xa.invert_Poisson(F32, ['lat', 'lon'], iParams=q)
        best = 1e9
        for _ in range(5):
            t = time.perf_counter()
            S = xa.invert_Poisson(F32, ['lat', 'lon'], iParams=q)
            best = min(best, time.perf_counter() - t)
        st = S.iParams['stats']
        print('invert_Poisson %dx%d, %d sweeps, %s: %.1f ms end to end (library call %.1f ms: h2d %.1f, sweeps %.1f, d2h %.1f)'
              % (nx, ny, iP['mxLoop'] + 1, tag, best * 1e3, st['wall_ms'], st['h2d_ms'], st['sweep_ms'], st['d2h_ms']))
    pr = cProfile.Profile()
    pr.enable()
    xa.invert_Poisson(F, ['lat', 'lon'], iParams=iP)
    pr.disable()
    pstats.Stats(pr).sort_stats('cumulative').print_stats(14)


if __name__ == '__main__':
    main()
