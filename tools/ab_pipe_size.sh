#!/bin/bash
# the size crossover between k_fused2d (K = 4) and the pipelined pass for the standard form, re-measured
for m in 1 2 3 4 8; do for mode in "XINV_PIPE=0" "XINV_PIPE=2 XINV_PIPE_FR=0" "XINV_PIPE=2 XINV_PIPE_FR=1"; do
  env $mode python tools/bench_configs.py c2 --members $m --reps 2 --sweeps 400 2>/dev/null | grep '^{' | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(sys.argv[1], sys.argv[2:], '%.4g' % d['point_sweeps_per_s'], {k:d[k] for k in d if k in ('rows_per_tile','sweeps_per_launch','masked_tile_pct','pipelined')})" $m $mode
done; done
