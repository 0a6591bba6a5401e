#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
( timeout 1200 python -m pytest tests/test_gpu_small.py tests/test_gpu_parity.py tests/test_gpu_frontend.py tests/test_gpu_plan.py -q -x -k "3d or pipe3d or tail or omega or 3D" 2>&1 | tail -4 )
for b in fixed extend; do for spl in 0 1; do python tools/bench_configs.py c5 --members 15 --bcy $b --spl $spl --reps 2 2>/dev/null | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('BCy $b', 'spl', d['sweeps_per_launch'], '%.4g' % d['point_sweeps_per_s'], 'launch %.1f us' % (d['avg_launch_ms'] * 1e3))"; done; done
