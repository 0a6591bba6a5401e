#!/bin/bash
# A/B harness: tools/ab.sh "<so list>" "<spl rows>" ... -- runs bench.py once per library variant and configuration
libs=$1; shift
for so in $libs; do
  for cfg in "$@"; do
    read -r spl rows <<< "$cfg"
    r=$(XINV_SO=$PWD/build/$so timeout 300 python bench.py --steps 3 --warmup 1 --sweeps 240 --no-cpu --spl $spl --rows $rows 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f G  launch %.1f us' % (d['value']/1e9, d['roofline']['avg_launch_ms']*1e3))")
    echo "$so spl=$spl rows=$rows: $r"
  done
done
