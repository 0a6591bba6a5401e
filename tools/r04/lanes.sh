cd "${GRAFT_REPO_ROOT:-$PWD}"; mkdir -p gpurun_out/lanes
run() { python tools/bench_configs.py "$@" --reps 3 2>/dev/null | grep '^{' | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('$TAG', d['config'], d['shape'], '%.4g  launch %.1f us' % (d['point_sweeps_per_s'], d['avg_launch_ms']*1e3))"; }
{
for m in 2 3 4 5 6; do
TAG=lag1 XINV_LANES=1 run c4 --members $m
TAG=lag2 XINV_LANES=2 run c4 --members $m
done
for m in 2 3; do
TAG=lag1 XINV_LANES=1 run c2 --members $m
TAG=lag2 XINV_LANES=2 run c2 --members $m
done
XINV_LANES=2 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_small.py tests/test_gpu_large.py tests/test_gpu_watchdog.py tests/test_gpu_lanes.py -q -x 2>&1 | tail -2
} > gpurun_out/lanes/out9.txt 2>&1
cat gpurun_out/lanes/out9.txt
