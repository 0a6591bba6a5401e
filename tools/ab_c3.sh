#!/bin/bash
# A/B of library builds on the k_fused2d configurations:  bash tools/ab_c3.sh so1 so2 ...   ("" = the shipped library)
for so in "$@"; do
  ( XINV_SO=$so python tools/bench_configs.py c3 --reps 3; XINV_SO=$so python tools/bench_configs.py c2 --no-xuniform --reps 3; XINV_PIPE=0 XINV_SO=$so python tools/bench_configs.py c2 c4 --reps 3 ) 2>/dev/null | grep '^{' | python -c "
import json,sys
print(sys.argv[1] or 'shipped', ' '.join('%s(K%d,um%d) %.4g' % (json.loads(l)['config'], json.loads(l)['sweeps_per_launch'], json.loads(l)['xuniform_mask'], json.loads(l)['point_sweeps_per_s']) for l in sys.stdin))" "$so"
done
