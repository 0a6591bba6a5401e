#!/bin/bash
# tile-height table for one config and member count:  bash tools/ab_rows_members.sh c3m 8 9 12 15 ...
cfg=$1; mem=$2; shift 2
for r in "$@"; do
  python tools/bench_configs.py $cfg --members $mem --rows $r --reps 2 2>/dev/null | grep '^{' | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('members', sys.argv[2], 'rows', sys.argv[1], '%.4g' % d['point_sweeps_per_s'], d['rows_per_tile'])" $r $mem
done
