#!/bin/bash
# A/B of two library builds over the configs:  bash tools/ab_so_configs.sh build/libxinv_x.so
for so in "" "$1" "" "$1"; do
  XINV_SO=$so python tools/bench_configs.py c1 c2 c3 c3m c4 c5 c5g --reps 3 2>/dev/null | grep '^{' | python -c "
import json,sys
print(sys.argv[1] or 'shipped', ' '.join('%s %.4g' % (json.loads(l)['config'], json.loads(l)['point_sweeps_per_s']) for l in sys.stdin))" "$so"
done
for so in "" "$1"; do
  XINV_SO=$so python tools/bench_ninepoint.py 2>/dev/null | grep '^{' | python -c "
import json,sys
print(sys.argv[1] or 'shipped', ' '.join('%.4g' % json.loads(l).get('point_sweeps_per_s', 0) for l in sys.stdin))" "$so"
  XINV_SO=$so python tools/bench_configs.py c4 --members 32 --reps 2 2>/dev/null | grep '^{' | python -c "
import json,sys
print(sys.argv[1] or 'shipped', ' '.join('%s32 %.4g' % (json.loads(l)['config'], json.loads(l)['point_sweeps_per_s']) for l in sys.stdin))" "$so"
  XINV_SO=$so python tools/bench_configs.py c5 --members 15 --reps 2 2>/dev/null | grep '^{' | python -c "
import json,sys
print(sys.argv[1] or 'shipped', ' '.join('%s15 %.4g' % (json.loads(l)['config'], json.loads(l)['point_sweeps_per_s']) for l in sys.stdin))" "$so"
done
