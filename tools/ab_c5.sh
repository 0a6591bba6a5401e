#!/bin/bash
# A/B of library builds on the 3-D configs:  bash tools/ab_c5.sh so1 so2 ...   ("" = the shipped library)
for so in "$@"; do
  ( XINV_SO=$so python tools/bench_configs.py c5 ofes --reps 3; XINV_SO=$so python tools/bench_configs.py c5 --members 15 --reps 3 ) 2>/dev/null | grep '^{' | python -c "
import json,sys
print(sys.argv[1] or 'shipped', ' '.join('%s%s %.4g' % (json.loads(l)['config'], json.loads(l)['shape'][0], json.loads(l)['point_sweeps_per_s']) for l in sys.stdin))" "$so"
done
