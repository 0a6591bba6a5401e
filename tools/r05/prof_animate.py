#!/usr/bin/env python3
"""cProfile of apps.animate_iteration as the reference's tests call it (73 x 144, 40 frames of 2 sweeps): where the Python side of a
call goes (profiles/r05_animate.txt)."""
import cProfile
import io
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import xinvert_amd as xa
from xinvert_amd import apps
lat = np.linspace(-90, 90, 73); lon = np.linspace(0, 360, 144, endpoint=False)
Q = 0.05 * np.exp(-((lat[:, None]) ** 2 + (lon[None, :] - 120.0) ** 2) / 100.0)
F = xa.Field(Q, ('lat', 'lon'), {'lat': lat, 'lon': lon})
def run():
    ip = {'BCs': ['fixed', 'periodic'], 'tolerance': 1e-12, 'optArg': 1.4}
    return apps.animate_iteration('GillMatsuno', F, dims=['lat', 'lon'], coords='lat-lon', mParams={'epsilon': 1e-5, 'Phi': 5000.0}, iParams=ip, loop_per_frame=2, max_frames=40)
for _ in range(3): run()
pr = cProfile.Profile(); pr.enable()
for _ in range(5): run()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(28); print(s.getvalue()[:4500])
