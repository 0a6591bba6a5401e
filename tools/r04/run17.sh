#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r04_run17
timeout 1200 python -m pytest tests/test_gpu_seam.py -q -x 2>&1 | tail -15 | tee gpurun_out/r04_run17/seam.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_small.py tests/test_gpu_scalar_cache.py tests/test_gpu_frontend.py -q -x 2>&1 | tail -5 | tee gpurun_out/r04_run17/parity.txt
python - <<'PY' 2>&1 | tee gpurun_out/r04_run17/seam_rates.txt
import time, numpy as np, torch
from xinvert_amd import synthetic
from xinvert_amd.resident import ResidentProblem
def rate(p, sweeps, **o):
    rp = ResidentProblem(p); best = 1e9
    for rep in range(3):
        rp.reset(); torch.cuda.synchronize(); t = time.perf_counter()
        fl, s = rp.solve(sweeps - 1, 0.0, timing=1, **o); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
    return rp.nb * rp.n * sweeps / best, s
for (ny, nx, nb) in ((1800, 3600, 1), (1801, 3601, 1), (1800, 3601, 1), (1800, 3600, 8), (1800, 3601, 8), (180, 360, 64), (180, 361, 64)):
    p = synthetic.poisson_latlon(ny, nx, mask=True, members=nb)
    v, s = rate(p, 400)
    print('poisson %dx%d x%d: %.4g  path %d pipelined %d K %d rows %d' % (ny, nx, nb, v, s['path'], s['pipelined'], s['sweeps_per_launch'], s['rows_per_tile']))
for (ny, nx, nb) in ((720, 1440, 8), (720, 1441, 8), (73, 144, 365), (73, 145, 365)):
    p = synthetic.gill_matsuno(ny, nx, nb)
    v, s = rate(p, 400)
    print('gill-matsuno %dx%d x%d: %.4g  path %d pipelined %d K %d' % (ny, nx, nb, v, s['path'], s['pipelined'], s['sweeps_per_launch']))
PY
