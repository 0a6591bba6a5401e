#!/usr/bin/env python3
"""apps.animate_iteration as the reference's tests use it (tests/test_AnimateConverge.py:13-31: Gill-Matsuno on 73 x 144,
40 frames of 2 sweeps): wall clock of the whole call on a resident plan (round 5) against the same call with every frame
re-deriving what a plan keeps (rounds 1-4: iParams['resident_plan'] = False).
  python tools/bench_animate.py [--frames 40] [--loops 2] [--ny 73 --nx 144]"""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.environ.get('XINV_TREE', ROOT))          # (XINV_TREE: another checkout of the package, e.g. a round-4 tree)
import xinvert_amd as xa
from xinvert_amd import apps

ap = argparse.ArgumentParser()
ap.add_argument('--frames', type=int, default=40); ap.add_argument('--loops', type=int, default=2)
ap.add_argument('--ny', type=int, default=73); ap.add_argument('--nx', type=int, default=144)
a = ap.parse_args()
lat = np.linspace(-90, 90, a.ny); lon = np.linspace(0, 360, a.nx, endpoint=False)
Q = 0.05 * np.exp(-((lat[:, None] - 0.0) ** 2 + (lon[None, :] - 120.0) ** 2) / 100.0)
F = xa.Field(Q, ('lat', 'lon'), {'lat': lat, 'lon': lon})
ref = None
for plan in (True, False, True, False):
    best = 1e9
    for rep in range(4):
        ip = {'BCs': ['fixed', 'periodic'], 'tolerance': 1e-12, 'optArg': 1.4, 'resident_plan': plan}
        t = time.perf_counter()
        out = apps.animate_iteration('GillMatsuno', F, dims=['lat', 'lon'], coords='lat-lon', mParams={'epsilon': 1e-5, 'Phi': 5000.0},
                                     iParams=ip, loop_per_frame=a.loops, max_frames=a.frames)
        best = min(best, time.perf_counter() - t)
    v = np.asarray(out.values)
    if ref is None:
        ref = v
    print(json.dumps({'resident_plan': plan, 'frames': a.frames, 'loops_per_frame': a.loops, 'shape': [a.ny, a.nx],
                      'wall_ms': best * 1e3, 'ms_per_frame': best * 1e3 / a.frames, 'planned': ip.get('stats', {}).get('planned', 0), 'tree': os.environ.get('XINV_TREE', 'this'),
                      'same_frames': bool(np.array_equal(v, ref))}), flush=True)
