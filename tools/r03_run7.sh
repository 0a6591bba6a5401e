#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_frontend.py tests/test_gpu_scalar_cache.py -m gpu -x -q ) > gpurun_out/r03_gputests_7.txt 2>&1
tail -6 gpurun_out/r03_gputests_7.txt
for fr in 0 1; do
  for mem in 8 64; do
    XINV_PIPE_FR=$fr python bench.py --config c4 --members $mem --steps 5 --warmup 2 --sweeps 200 2>&1 | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); r = d['roofline']
        print('FR=$fr c4 members $mem value %.4g  launch %.1f us  K %d rows %d' % (d['value'], r['avg_launch_ms']*1e3, d['config']['sweeps_per_launch'], d['config']['rows_per_tile']))
"
  done
  XINV_PIPE_FR=$fr python bench.py --steps 10 --warmup 3 --no-cpu --no-parity --no-hbm --no-configs 2>&1 | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); r = d['roofline']
        print('FR=$fr c2 headline value %.4g  launch %.2f us' % (d['value'], r['avg_launch_ms']*1e3))
"
done 2>&1 | tee gpurun_out/r03_fr_ab.txt
python tools/bench_host_pipeline.py c5 --members 15 --sweeps 200 2>&1 | grep -v amdgpu | head -3 | cut -c1-200
