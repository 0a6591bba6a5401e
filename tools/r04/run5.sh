#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r04_run5
timeout 1500 python -m pytest tests/test_gpu_fma.py -q -x 2>&1 | tail -15 | tee gpurun_out/r04_run5/fma.txt
python - <<'PY' 2>&1 | tee gpurun_out/r04_run5/fma_rates.txt
import time, numpy as np, torch
from xinvert_amd import synthetic
from xinvert_amd.resident import ResidentProblem
def rate(p, sweeps, **o):
    rp = ResidentProblem(p); best = 1e9
    for rep in range(4):
        rp.reset(); torch.cuda.synchronize(); t = time.perf_counter()
        fl, s = rp.solve(sweeps - 1, 0.0, timing=1, **o); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
    return rp.nb * rp.n * sweeps / best, s['sweep_ms'] / s['sweep_launches'] * 1e3
for name, p, sw in (('C2 poisson 1800x3600', synthetic.poisson_latlon(1800, 3600, mask=True), 500),
                    ('C2 x8', synthetic.poisson_latlon(1800, 3600, mask=True, members=8), 200),
                    ('C4 gill-matsuno 720x1440 x8', synthetic.gill_matsuno(720, 1440, 8), 500),
                    ('C5 omega 50x360x720 x4', synthetic.omega_latlon(50, 360, 720, steps=4), 100)):
    for o in (dict(), dict(fma=1), dict(no_tile_skip=1), dict(no_tile_skip=1, fma=1)):
        v, us = rate(p, sw, **o)
        print('%-30s %-32r %.4g  launch %.1f us' % (name, o, v, us), flush=True)
PY
