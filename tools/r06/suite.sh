#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06_suite
( time timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -15 ) 2>&1 | tee gpurun_out/r06_suite/gputests.txt
for r in 0 9 15 30 60; do python tools/bench_configs.py c3m c3mxy --rows $r --reps 3 2>/dev/null | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['config'], 'rows', d['rows_per_tile'], '%.4g' % d['point_sweeps_per_s'], 'launch %.1f us' % (d['avg_launch_ms'] * 1e3))"; done | tee gpurun_out/r06_suite/munk_xy.txt
python tools/bench_configs.py c3mxy --path 1 --reps 2 2>/dev/null | grep '^{' | cut -c1-200 | tee -a gpurun_out/r06_suite/munk_xy.txt
