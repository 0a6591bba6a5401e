#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_frontend.py -m gpu -x -q ) > gpurun_out/r03_gputests_6.txt 2>&1
tail -6 gpurun_out/r03_gputests_6.txt
python - <<'PY'
import os, sys, time, json
sys.path.insert(0, os.getcwd())
import numpy as np
from xinvert_amd import synthetic
from xinvert_amd.resident import ResidentProblem
import torch
def run(name, p, sweeps, **o):
    rp = ResidentProblem(p)
    for rep in range(2):
        rp.reset(); torch.cuda.synchronize(); t = time.perf_counter()
        fl, s = rp.solve(sweeps - 1, 0.0, timing=1, **o)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
    print('%-28s %-18s value %.4g  launch %.1f us  K %d rows %d um %d pipelined %d' % (name, o, rp.nb * rp.n * sweeps / dt, s['sweep_ms'] / s['sweep_launches'] * 1e3, s['sweeps_per_launch'], s['rows_per_tile'], s['xuniform_mask'], s['pipelined']), flush=True)
st = synthetic.stommel_cartesian(2000, 2000)
for o in (dict(), dict(no_pipe=1)):
    run('C3 Stommel 2000x2000', st, 300, **o)
c2 = synthetic.poisson_latlon(1800, 3600, mask=True)
for o in (dict(no_xuniform=1), dict(no_xuniform=1, no_pipe=1), dict(no_xuniform=1, no_tile_skip=1), dict(no_xuniform=1, no_tile_skip=1, no_pipe=1)):
    run('C2 all arrays streamed', c2, 300, **o)
PY
