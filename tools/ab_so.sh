#!/bin/bash
# A/B of library builds on one config:  bash tools/ab_so.sh c3m "rows..." so1 so2 ...   ("" = the shipped library)
cfg=$1; rows=$2; shift 2
for so in "$@"; do for r in $rows; do
  XINV_SO=$so python tools/bench_configs.py $cfg --rows $r --reps 3 2>/dev/null | grep '^{' | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(sys.argv[2] or 'shipped', 'rows', sys.argv[1], '%.4g' % d['point_sweeps_per_s'])" $r "$so"
done; done
