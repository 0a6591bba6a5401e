cd "${GRAFT_REPO_ROOT:-$PWD}"; mkdir -p gpurun_out/seam3d
run() { python tools/bench_configs.py "$@" --reps 3 2>/dev/null | grep '^{' | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('$TAG', d['config'], d['shape'], '%.4g  launch %.1f us' % (d['point_sweeps_per_s'], d['avg_launch_ms']*1e3))"; }
{
TAG=even_two_sweeps run c5 --members 15
TAG=even_one_sweep run c5 --members 15 --spl 1
TAG=odd_seam run c5odd --members 15
TAG=odd_colour run c5odd --members 15 --path 1
TAG=even_one_sweep run c5 --members 2 --spl 1
TAG=odd_seam run c5odd --members 2
TAG=odd_colour run c5odd --members 2 --path 1
( time timeout 2400 python -m pytest tests -m gpu -q -x ) 2>&1 | tail -5
} > gpurun_out/seam3d/out.txt 2>&1
cat gpurun_out/seam3d/out.txt
