#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for r in 0 -20 -27 -33 -40 -50 -60 -80; do python tools/bench_configs.py c2 --rows $r --reps 3 --sweeps 500 2>/dev/null | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('rows opt', '$r', 'rows', d['rows_per_tile'], '%.4g' % d['point_sweeps_per_s'], 'launch %.2f us' % (d['avg_launch_ms'] * 1e3), 'masked', d['masked_tile_pct'])"; done
