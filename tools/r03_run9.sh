#!/bin/bash
# round 3: A/B of the EXEC-masked update / norm share in k_pipe2d (XINV_PIPE_EXECSEL) + the pipelined-pass parity tests
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q ) > gpurun_out/r03_gputests_9.txt 2>&1
tail -4 gpurun_out/r03_gputests_9.txt
cat > /tmp/ab.py <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
from xinvert_amd import synthetic
from xinvert_amd.resident import ResidentProblem
import torch
def run(name, p, sweeps, **o):
    rp = ResidentProblem(p)
    for rep in range(3):
        rp.reset(); torch.cuda.synchronize(); t = time.perf_counter()
        fl, s = rp.solve(sweeps - 1, 0.0, timing=1, **o)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
    print('%s %-24s %-22s value %.4g  launch %.1f us  K %d rows %d pipe %d' % (os.environ.get('XINV_SO', 'main')[-16:], name, o, rp.nb * rp.n * sweeps / dt, s['sweep_ms'] / s['sweep_launches'] * 1e3, s['sweeps_per_launch'], s['rows_per_tile'], s.get('pipelined', -1)), flush=True)
run('C2 3600x1800', synthetic.poisson_latlon(1800, 3600, mask=True), 500)
run('C2 3600x1800 nomask', synthetic.poisson_latlon(1800, 3600, mask=False), 500)
run('C1 360x180', synthetic.poisson_latlon(180, 360, mask=False), 500)
run('C4 8', synthetic.gill_matsuno(720, 1440, 8), 200)
run('C4 64', synthetic.gill_matsuno(720, 1440, 64), 200)
PY
for so in "" build/libxinv_noexec.so ""  build/libxinv_noexec.so; do XINV_SO=$so python /tmp/ab.py; done 2>&1 | grep -v amdgpu | tee gpurun_out/r03_execsel_ab.txt
