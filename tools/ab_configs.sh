#!/bin/bash
# A/B over the BASELINE configurations: tools/ab_configs.sh "<so list>" c1 c3 ... [-- extra bench_configs args]
libs=$1; shift
for so in $libs; do
  XINV_SO=$PWD/build/$so timeout 600 python tools/bench_configs.py "$@" 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if not l.startswith('{'): continue
    d = json.loads(l)
    print('$so %-5s %7.1f G  launch %7.1f us' % (d.get('config', '?'), d['point_sweeps_per_s'] / 1e9, d['avg_launch_ms'] * 1e3))
"
done
