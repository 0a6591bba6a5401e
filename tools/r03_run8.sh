#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 1800 python -m pytest tests -m gpu -x -q ) > gpurun_out/r03_gputests_8.txt 2>&1
tail -6 gpurun_out/r03_gputests_8.txt
cat > /tmp/rq_ab.py <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
from xinvert_amd import synthetic
from xinvert_amd.resident import ResidentProblem
import torch
def run(name, p, sweeps, **o):
    rp = ResidentProblem(p)
    for rep in range(2):
        rp.reset(); torch.cuda.synchronize(); t = time.perf_counter()
        fl, s = rp.solve(sweeps - 1, 0.0, timing=1, **o)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
    print('RQPRE=%s %-24s %-22s value %.4g  launch %.1f us  K %d rows %d um %d' % (os.environ.get('XINV_RQPRE', '1'), name, o, rp.nb * rp.n * sweeps / dt, s['sweep_ms'] / s['sweep_launches'] * 1e3, s['sweeps_per_launch'], s['rows_per_tile'], s['xuniform_mask']), flush=True)
st = synthetic.stommel_cartesian(2000, 2000)
for o in (dict(), dict(sweeps_per_launch=2), dict(sweeps_per_launch=1)):
    run('C3 Stommel 2000x2000', st, 300, **o)
c2 = synthetic.poisson_latlon(1800, 3600, mask=True)
for o in (dict(no_xuniform=1), dict(no_xuniform=1, sweeps_per_launch=2)):
    run('C2 all arrays streamed', c2, 300, **o)
PY
for q in 1 0; do XINV_RQPRE=$q python /tmp/rq_ab.py; done 2>&1 | grep -v amdgpu | tee gpurun_out/r03_rq_ab.txt
