#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06_bih
( timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -k "biharmonic" 2>&1 | tail -3 ) | tee gpurun_out/r06_bih/tests.txt
for r in 0 15 21 27 33; do python tools/bench_configs.py c3mxy --rows $r --reps 3 2>/dev/null | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['config'], 'rows', d['rows_per_tile'], '%.4g' % d['point_sweeps_per_s'], 'launch %.1f us' % (d['avg_launch_ms'] * 1e3), 'vm', d.get('point_factor'))"; done | tee gpurun_out/r06_bih/munk_xy_vm3.txt
