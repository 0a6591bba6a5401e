cd "${GRAFT_REPO_ROOT:-$PWD}"; mkdir -p gpurun_out/lanes
run() { python tools/bench_configs.py "$@" --reps 3 2>/dev/null | grep '^{' | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('$TAG', d['config'], d['shape'], '%.4g  launch %.1f us' % (d['point_sweeps_per_s'], d['avg_launch_ms']*1e3))"; }
{
for cfg in "c1 --members 4" "c1 --members 8" "c1 --members 16" "c1 --members 32" "c1 --members 64" "c4 --members 2" "c4 --members 3" "c5 --members 2" "c5 --members 3" "c5 --members 6" "c5 --members 8"; do
for n in 1 2; do TAG=lanes$n XINV_LANES=$n run $cfg; done
done
} > gpurun_out/lanes/out5.txt 2>&1
cat gpurun_out/lanes/out5.txt
