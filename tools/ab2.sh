#!/bin/bash
# A/B harness: tools/ab2.sh "<tags>" -- headline + configs with each library variant build/libxinv_<tag>.so ("cur" = in-tree)
for tag in $1; do
  so=$PWD/build/libxinv_$tag.so; [ "$tag" = cur ] && so=$PWD/xinvert_amd/libxinv_hip.so
  r=$(XINV_SO=$so timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu --no-parity --no-hbm 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f G  launch %.1f us  K=%d RY=%d' % (d['value']/1e9, d['roofline']['avg_launch_ms']*1e3, d['config']['sweeps_per_launch'], d['config']['rows_per_tile']))")
  echo "$tag c2: $r"
  XINV_SO=$so timeout 600 python tools/bench_configs.py ${2:-c1 c3 c4} 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('$tag', d['config'], d['shape'], '%.1f G' % (d['point_sweeps_per_s']/1e9), 'launch %.1f us' % (d['avg_launch_ms']*1e3), 'K', d['sweeps_per_launch'], 'RY', d['rows_per_tile'])
"
done
