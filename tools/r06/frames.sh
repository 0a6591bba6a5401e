#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
python tools/bench_animate.py 2>&1 | grep "^{\|Error\|error" | cut -c1-250
python - <<'PY'
import cProfile, pstats, sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
import xinvert_amd as xa
from xinvert_amd import apps
lat = np.linspace(-90, 90, 73); lon = np.linspace(0, 360, 144, endpoint=False)
Q = 0.05 * np.exp(-((lat[:, None] - 0.0) ** 2 + (lon[None, :] - 120.0) ** 2) / 100.0)
F = xa.Field(Q, ('lat', 'lon'), {'lat': lat, 'lon': lon})
def run():
    ip = {'BCs': ['fixed', 'periodic'], 'tolerance': 1e-12, 'optArg': 1.4}
    return apps.animate_iteration('GillMatsuno', F, dims=['lat', 'lon'], coords='lat-lon', mParams={'epsilon': 1e-5, 'Phi': 5000.0}, iParams=ip, loop_per_frame=2, max_frames=40)
for _ in range(5): run()
pr = cProfile.Profile(); pr.enable()
for _ in range(20): run()
pr.disable()
st = pstats.Stats(pr); st.sort_stats('cumulative').print_stats(28)
PY
