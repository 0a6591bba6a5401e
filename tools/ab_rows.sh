#!/bin/bash
# A/B of the tile height for one config:  bash tools/ab_rows.sh c3m 0 9 12 ...
cfg=$1; shift
for r in "$@"; do
  python tools/bench_configs.py $cfg --rows $r --reps 2 2>/dev/null | grep '^{' | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('rows', sys.argv[1], '%.4g' % d['point_sweeps_per_s'], {k:d[k] for k in d if k in ('rows_per_tile','sweeps_per_launch','masked_tile_pct')})" $r
done
