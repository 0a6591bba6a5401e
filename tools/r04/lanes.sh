cd "${GRAFT_REPO_ROOT:-$PWD}"; mkdir -p gpurun_out/lanes
run() { python tools/bench_configs.py "$@" --reps 3 2>/dev/null | grep '^{' | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('$TAG', d['config'], d['shape'], '%.4g  launch %.1f us' % (d['point_sweeps_per_s'], d['avg_launch_ms']*1e3))"; }
{
for rep in 1 2 3; do
TAG=one XINV_LANES=1 run c5 --members 15
TAG=split6 XINV_LANES=2 XINV_LANE0=6 run c5 --members 15
TAG=split7 XINV_LANES=2 XINV_LANE0=7 run c5 --members 15
done
for m in 10 12 16 20 24; do
TAG=one XINV_LANES=1 run c5 --members $m
TAG=split40 XINV_LANES=2 XINV_LANE0=$((m*2/5)) run c5 --members $m
done
} > gpurun_out/lanes/out7.txt 2>&1
cat gpurun_out/lanes/out7.txt
