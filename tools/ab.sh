#!/bin/bash
# A/B harness: tools/ab.sh "<so list>" "<spl rows> ..." -- runs bench.py per library variant
for so in $1; do
  for cfg in "${@:2}"; do set -- $cfg
    r=$(XINV_SO=$PWD/build/$so timeout 300 python bench.py --steps 3 --warmup 1 --sweeps 240 --no-cpu --spl $1 --rows $2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f G  launch %.1f us' % (d['value']/1e9, d['roofline']['avg_launch_ms']*1e3))")
    echo "$so spl=$1 rows=$2: $r"
  done
done
